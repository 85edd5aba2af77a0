import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from pion_amd import abi, problems, lib, driver
from cpu_backends import CpuSim
sv=int(sys.argv[1]) if len(sys.argv)>1 else 5
cfg,P=problems.hd_blast_octant(20,3,solver=sv,ntracer=1,strict_fp=1,nzones=3.0)
g=lib.GpuSim(cfg,0); o=CpuSim(cfg,'orc')
sg,so=driver.SimControl(g,cfg),driver.SimControl(o,cfg)
sg.init(P); so.init(P)
sg.calculate_timestep(); so.calculate_timestep()
g.stage(0.5*sg.dt,1,0); o.stage(0.5*so.dt,1,0)
a,b=g.download(1)[:,2:-2,2:-2,2:-2],o.download(1)[:,2:-2,2:-2,2:-2]
p0=P[:,2:-2,2:-2,2:-2]
for v in range(cfg.nvar):
    d=(a[v]!=b[v]); print("var",v,"ndiff", d.sum())
    for i in np.argwhere(d)[:30]:
        i=tuple(i); print("  zyx",i,"gpu",a[v][i],"orc",b[v][i],"init",p0[v][i])
