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
for sv in (0,1,3,4,5,6,8):
    cfg,P=problems.hd_blast_octant(20,3,solver=sv,ntracer=1,strict_fp=1,nzones=3.0)
    run(cfg,P,"HD solver %d ntr1"%sv)
import copy
for eq in (abi.EQMHD,abi.EQGLM):
  for sv in (7,8):
    for ntr in (1,2):
        cfg,P0=problems.mhd_blastwave(16,3,eq,sv,strict_fp=1)
        base=cfg.nvar
        cfg=abi.make_config(3,[16,16,16],eq,sv,ntracer=ntr,etav=0.1,gamma=5/3,cfl=0.24,xmin=(-.5,-.5,-.5),xmax=(.5,.5,.5),refvec=[1,0.1]+[1]*14,strict_fp=1)
        P=np.zeros((cfg.nvar,)+P0.shape[1:]); P[:base]=P0
        for t in range(ntr): P[base+t]=(P0[1]>1)*1.0
        run(cfg,P,"eq %d solver %d ntr %d"%(eq,sv,ntr))
