import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from pion_amd import abi, problems, lib, driver
from cpu_backends import CpuSim
def run(cfg,P,label):
    g=lib.GpuSim(cfg,0); o=CpuSim(cfg,'orc')
    sg,so=driver.SimControl(g,cfg),driver.SimControl(o,cfg)
    sg.init(P); so.init(P)
    sg.calculate_timestep(); so.calculate_timestep()
    g.stage(0.5*sg.dt,1,0); o.stage(0.5*so.dt,1,0)
    a,b=g.download(1)[:,2:-2,2:-2,2:-2],o.download(1)[:,2:-2,2:-2,2:-2]
    out=[]
    for v in range(cfg.nvar):
        sc=np.abs(b[v]).max()+1e-300
        out.append("%.1e"%(np.abs(a[v]-b[v]).max()/sc))
    print(label,out)
for sv in (5,6):
  for av in (0,1):
    cfg,P=problems.hd_blast_octant(20,3,solver=sv,ntracer=1,artvisc=av,strict_fp=1,nzones=3.0)
    run(cfg,P,"HD solver %d ntr1 av%d"%(sv,av))
