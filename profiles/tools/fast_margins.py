"""How far inside its gates the fast (benchmarked) build sits: the long reference runs of tests/test_endstate.py
(L1 / L2 of the end state over refvec, conserved totals) and the 64^3 ten-step blast of tests/test_gpu_xtile.py.
    python profiles/tools/fast_margins.py            (PION_GPU_LIB picks the library)
Gate for every number: 1e-10."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as gc  # noqa: E402
from pion_amd import lib  # noqa: E402

gold = np.load(os.path.join(ROOT, "tests", "golden", "endstate.npz"))
print("library:", os.environ.get("PION_GPU_LIB", "in-tree"))
for name in gc.END_CASES:
    cfg, P, tf, nmax = gc.end_case(name, strict_fp=0)
    with lib.GpuSim(cfg, 0) as g:
        n, t, dts = gc.end_run(g, cfg, P, tf, nmax)
        A = g.download(0)
    tot, mag = gc.conserved_totals(cfg, A)
    rel = np.abs(tot - gold[name + "_tot"]) / (mag + 1e-300)
    l1, l2, mx = gc.diff_norms(cfg, A, gold[name + "_P"])
    print("%-22s steps %4d (ref %4d)  dt rel %.2e  L1 %.2e  L2 %.2e  max %.2e  totals %.2e" % (
        name, n, int(gold[name + "_n"]), np.abs(dts[:min(len(dts), len(gold[name + "_dt"]))] / gold[name + "_dt"][:len(dts)] - 1).max(),
        l1.max(), l2.max(), mx.max(), rel.max()))
