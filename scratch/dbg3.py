import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from pion_amd import abi, problems, lib, driver
from cpu_backends import CpuSim
def run(sv,ntr,strict,ooa_stage):
    cfg,P=problems.hd_blast_octant(20,3,solver=sv,ntracer=ntr,strict_fp=strict,nzones=3.0)
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
    print("solver",sv,"ntr",ntr,"strict",strict,out)
for sv in (4,5,6,8):
    for ntr in (0,1,2):
        for strict in (1,0):
            run(sv,ntr,strict,1)
