import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from pion_amd import abi, driver, problems, lib
from cpu_backends import CpuSim
eq, solver = int(sys.argv[1]), int(sys.argv[2])
def run(strict, nsteps, cpu=False):
    cfg, P = problems.mhd_blastwave(14, 3, eq, solver, strict_fp=strict)
    with (CpuSim(cfg, "orc") if cpu else lib.GpuSim(cfg, 0)) as g:
        sc = driver.SimControl(g, cfg); sc.init(P)
        out = []
        for _ in range(nsteps):
            sc.calculate_timestep(); sc.advance_time(); out.append(g.download(0))
        return out
ref = run(1, 3, cpu=True)
res = {}
for k in ("rows", "march", "cell"):
    os.environ["PION_STAGE_KERNEL"] = k
    res[k] = run(0, 3)
    for it in range(3):
        d = np.abs(res[k][it] - ref[it]); sc = np.abs(ref[it]).reshape(ref[it].shape[0], -1).max(axis=1)
        rel = d.reshape(d.shape[0], -1).max(axis=1) / (sc + 1e-300)
        w = np.unravel_index(np.argmax(d / sc.reshape(-1, 1, 1, 1)), d.shape)
        print(k, "step", it, "max rel per var", " ".join("%.1e" % x for x in rel), "at", w)
for it in range(3):
    print("rows-vs-cell", it, np.abs(res["rows"][it] - res["cell"][it]).max(), "march-vs-cell", np.abs(res["march"][it] - res["cell"][it]).max())
import torch
print("torch", torch.__version__, torch.cuda.is_available(), torch.cuda.device_count())
