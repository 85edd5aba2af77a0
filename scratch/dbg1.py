import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from pion_amd import abi, problems, lib
from cpu_backends import CpuSim
rng = np.random.default_rng(7)
eq=abi.EQEUL
for sv in (1,):
    cfg = abi.make_config(3, [4, 4, 4], eq, sv, ntracer=0, artvisc=1, xmax=(1, 1, 1), strict_fp=1)
    L, R = problems.random_states(rng, 2000, eq, 0)
    aux=np.zeros((2000,4))
    g=lib.GpuSim(cfg,0); o=CpuSim(cfg,'orc')
    Fg,Pg=g.interface_flux(0,L,R,aux,dt=0.01); Fo,Po=o.interface_flux(0,L,R,aux,dt=0.01)
    bad=np.argwhere(Fg!=Fo)
    print("nbad",len(bad))
    for i,v in bad[:6]:
        print(i,v,repr(Fg[i,v]),repr(Fo[i,v]),"L",L[i],"R",R[i], "Pstar g",Pg[i],"o",Po[i])
