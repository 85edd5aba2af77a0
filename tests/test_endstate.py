"""Long runs: low-resolution versions of BASELINE configs 1-4 against the END STATE of the reference's own
solver objects (tests/golden/endstate.npz, made by tests/golden/make_golden.py c from oracle/_ref):
171-400 steps each, shocks crossing most of the grid.

CPU part (this container and the GPU box): the oracle reproduces every dt and the end state bit for bit.
GPU part: the strict build does the same through the C-ABI (the hybrid Riemann solver of config 1 calls
exp / log / pow: device libm, <= 1e-9); the fast (benchmarked) build is held to the SURVEY 8(d) gates:
per-variable L1 and L2 of the difference <= 1e-10 x refvec, conserved totals within 1e-10 relative."""
import os

import numpy as np
import pytest

import golden_cases as gc
from pion_amd import abi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "endstate.npz")


@pytest.fixture(scope="module")
def gold():
    z = np.load(GOLD)
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", gc.END_CASES)
def test_oracle_reproduces_reference_end_state(gold, name):
    from cpu_backends import CpuSim
    cfg, P, tf, nmax = gc.end_case(name)
    with CpuSim(cfg, "orc") as o:
        n, t, dts = gc.end_run(o, cfg, P, tf, nmax)
        A = o.download(0)
    assert n == int(gold[name + "_n"]) and t == float(gold[name + "_t"])
    assert np.array_equal(dts, gold[name + "_dt"])
    assert np.array_equal(A, gold[name + "_P"]), name
    tot, _ = gc.conserved_totals(cfg, A)
    assert np.array_equal(tot, gold[name + "_tot"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", gc.END_CASES)
def test_gpu_strict_reproduces_reference_end_state(gold, name):
    from pion_amd import lib
    cfg, P, tf, nmax = gc.end_case(name, strict_fp=1)
    with lib.GpuSim(cfg, 0) as g:
        n, t, dts = gc.end_run(g, cfg, P, tf, nmax)
        A = g.download(0)
    assert n == int(gold[name + "_n"])
    if cfg.solver in (abi.FLUX_RSlinear, abi.FLUX_RSexact, abi.FLUX_RShybrid):
        # exp / log / pow of the device maths library differ from glibc in the last bits
        assert np.allclose(dts, gold[name + "_dt"], rtol=1e-9, atol=0.0)
        l1, l2, mx = gc.diff_norms(cfg, A, gold[name + "_P"])
        assert l1.max() <= 1e-9 and l2.max() <= 1e-9, (l1, l2, mx)
    else:
        assert t == float(gold[name + "_t"])
        assert np.array_equal(dts, gold[name + "_dt"])
        assert np.array_equal(A, gold[name + "_P"]), name


# what the fast build is held to per case: (L1 and L2 / refvec, conserved totals relative to the sum of
# magnitudes).  The SURVEY 8(d) gate is 1e-10 for both; where a case needs more the reason is given.
FAST_TOL = {
    "sph1d_n128": (1e-10, 1e-10),
    "dmr_n065": (1e-10, 1e-10),
    "mhd_bw2d_64x96": (1e-10, 1e-10),
    # the symmetric blast in ideal MHD is a discontinuity of the reference algorithm itself (1 ulp in ->
    # 2e-4 out after two steps, tests/test_reference_conditioning.py): no build-independent end state exists
    # cell by cell; the conserved totals do not care which branch a degenerate interface took
    "mhd_ideal_bw2d_64x96": (None, 1e-10),
    "bw3d_nr032": (1e-10, 1e-10),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", gc.END_CASES)
def test_gpu_fast_build_end_state_norms(gold, name):
    from pion_amd import lib
    cfg, P, tf, nmax = gc.end_case(name, strict_fp=0)
    with lib.GpuSim(cfg, 0) as g:
        n, t, dts = gc.end_run(g, cfg, P, tf, nmax)
        A = g.download(0)
    assert np.isfinite(A).all()
    assert abs(n - int(gold[name + "_n"])) <= 1
    tol_norm, tol_tot = FAST_TOL[name]
    tot, mag = gc.conserved_totals(cfg, A)
    if n == int(gold[name + "_n"]) and abs(t - float(gold[name + "_t"])) <= 1e-9 * abs(t):
        rel = np.abs(tot - gold[name + "_tot"]) / (mag + 1e-300)
        assert rel.max() <= tol_tot, rel
        if tol_norm is not None:
            l1, l2, mx = gc.diff_norms(cfg, A, gold[name + "_P"])
            assert l1.max() <= tol_norm and l2.max() <= tol_norm, (name, l1, l2, mx)
    else:
        pytest.fail("fast build took %d steps to t = %.17g, reference %d to %.17g" % (
            n, t, int(gold[name + "_n"]), float(gold[name + "_t"])))
