import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
order = sys.argv[1]
if order == "torch_first":
    import torch
    print("torch first: avail", torch.cuda.is_available())
    x = torch.ones(4, device="cuda:0", dtype=torch.float64)
import numpy as np
from pion_amd import abi, driver, problems, lib
cfg, P = problems.mhd_blastwave(14, 3, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=1)
with lib.GpuSim(cfg, 0) as g:
    sc = driver.SimControl(g, cfg); sc.init(P)
    sc.calculate_timestep(); sc.advance_time()
    a = g.download(0)
    print("sim ok", float(a.sum()))
    import torch
    print("avail after lib", torch.cuda.is_available())
    try:
        y = torch.ones(4, device="cuda:0", dtype=torch.float64) * 2
        print("tensor ok", y.sum().item())
        s = torch.cuda.Stream()
        g.set_stream(s.cuda_stream)
        sc.calculate_timestep(); sc.advance_time()
        print("stream ok", float(g.download(0).sum()))
    except Exception as e:
        print("EXC", repr(e)[:200])
os.system("grep -E 'amdhip64|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
