"""round-1 junk-output reproducer: strict k_stage_rows<MHD,1,HLL> of the round-1 sources, variant libraries"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pion_amd import abi, driver, problems, lib
from cpu_backends import CpuSim
from test_gpu_parity import _with_tracers
for eq, solver, ntr in ((abi.EQMHD, 8, 1), (abi.EQMHD, 8, 2), (abi.EQGLM, 8, 2), (abi.EQMHD, 7, 1)):
    cfg0, P0 = problems.mhd_blastwave(14, 3, eq, solver, strict_fp=1)
    cfg, P = _with_tracers(cfg0, P0, ntr)
    try:
        with lib.GpuSim(cfg, 0) as g, CpuSim(cfg, "orc") as o:
            sg, so = driver.SimControl(g, cfg), driver.SimControl(o, cfg)
            sg.init(P); so.init(P)
            sg.calculate_timestep(); so.calculate_timestep(); so.dt = sg.dt
            sg.advance_time(); so.advance_time()
            a, b = g.download(0), o.download(0)
    except Exception as e:
        print(os.environ.get("PION_GPU_LIB", "default").split("/")[-1], "eq", eq, "solver", solver, "ntr", ntr, "FAILED:", str(e)[:90])
        continue
    bad = (a != b)
    print(os.environ.get("PION_GPU_LIB", "default").split("/")[-1], "eq", eq, "solver", solver, "ntr", ntr,
          "differing values", int(bad.sum()), "per var", [int(bad[v].sum()) for v in range(cfg.nvar)],
          "sample", a[bad][:4] if bad.any() else "")
