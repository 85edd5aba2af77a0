import sys, time, copy
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import numpy as np
from pion_amd import abi, driver, problems, lib
from test_gpu_split_stage import SelfComm
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg, _ = problems.mhd_blastwave(4, 3, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=0)
cfg.ng[0] = cfg.ng[1] = 512; cfg.ng[2] = nz; cfg.dx = 1.0 / 512
cfg.xmin[2] = -0.5 * nz / 512
P = problems.fill_mhd_blastwave(cfg)
def run(mode):
    c = copy.deepcopy(cfg)
    if mode != "whole":
        c.bc_type[4] = c.bc_type[5] = abi.BC_SLAB
    with lib.GpuSim(c, 0) as g:
        comm = None if mode == "whole" else SelfComm(g, mode == "streams")
        sc = driver.SimControl(g, c, comm=comm)
        sc.init(P)
        for _ in range(2):
            sc.calculate_timestep(); sc.advance_time()
        sc.finish_halo(); g.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            sc.calculate_timestep(); sc.advance_time()
        sc.finish_halo(); g.synchronize(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
for mode in ("whole", "one_stream", "streams"):
    print(mode, "ms/step %.3f" % run(mode))
