import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from pion_amd import abi, driver, problems, lib
from cpu_backends import CpuSim
from test_gpu_parity import _with_tracers
eq, solver, ntr = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg0, P0 = problems.mhd_blastwave(14, 3, eq, solver, strict_fp=1)
cfg, P = _with_tracers(cfg0, P0, ntr)
with lib.GpuSim(cfg, 0) as g, CpuSim(cfg, "orc") as o:
    sg, so = driver.SimControl(g, cfg), driver.SimControl(o, cfg)
    sg.init(P); so.init(P)
    for it in range(2):
        dg, do = sg.calculate_timestep(), so.calculate_timestep()
        print("dt", dg, do)
        so.dt = sg.dt
        # stage by stage
        dt = sg.dt
        for (d, ooa, full, cs) in ((0.5 * dt, 1, 0, 1), (dt, 2, 1, 2)):
            g.stage(d, ooa, full); o.stage(d, ooa, full)
            a, b = g.download(1 - full), o.download(1 - full)
            nb = cfg.nbc
            for v in range(cfg.nvar):
                d_ = (a[v] != b[v])
                if d_.any():
                    idx = np.argwhere(d_)
                    print("it", it, "stage", cs, "var", v, "ndiff", d_.sum(), "first", idx[0], a[v][tuple(idx[0])], b[v][tuple(idx[0])],
                          "z range", idx[:, 0].min(), idx[:, 0].max(), "y", idx[:, 1].min(), idx[:, 1].max(), "x", idx[:, 2].min(), idx[:, 2].max())
            sg.update_bcs(cs, 2); so.update_bcs(cs, 2)
        sg.simtime += dt; so.simtime += dt; sg.last_dt = so.last_dt = dt; sg.timestep += 1; so.timestep += 1
print("done")
