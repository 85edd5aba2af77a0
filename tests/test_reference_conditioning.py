"""Is the REFERENCE itself continuous on the states the fast build is compared on?  (CPU, oracle / _ref)

Round-1 finding: on the Stone MHD blast (v = 0, B_z = 0 exactly) the fast (FMA) build and the strict
build of the ideal-MHD HLLD path drift apart by ~7e-4 within two steps.  This file shows that the drift is
a property of the reference algorithm on that degenerate input, not of the fast build: perturbing the
oracle's OWN input by +-1 ulp changes its result by the same order.  Mechanism: the HLLD -> HLL switch
tests `div v < 0` (solver_eqn_mhd_adi.cpp:167-181) where div v is rounding noise by symmetry, and for
B_n -> +-0 the U** states of HLLD_MHD.cpp:912-942 flip with sgn(B_n) while S*_L, S_M, S*_R coincide.  The
plain HLL solver on the same input, GLM-MHD on the same input, and ideal MHD on a blast with a velocity
field of definite divergence and B_z != 0 (problems.mhd_blast_generic) are all well conditioned
(1 ulp -> 1e-15), and the latter is where the fast build is compared with the oracle cell by cell
(tests/test_gpu_xtile.py)."""
import numpy as np
import pytest

from pion_amd import abi, driver, problems
from cpu_backends import CpuSim, have_ref


def _run(cfg, P, nsteps, kind):
    with CpuSim(cfg, kind) as o:
        sc = driver.SimControl(o, cfg)
        sc.init(P)
        for _ in range(nsteps):
            sc.calculate_timestep()
            sc.advance_time()
        return o.download(0).copy()


def _one_ulp(P, variables, seed=1):
    rng = np.random.default_rng(seed)
    Q = P.copy()
    for v in variables:
        up = rng.integers(0, 2, Q[v].shape) > 0
        Q[v] = np.where(up, np.nextafter(Q[v], np.inf), np.nextafter(Q[v], -np.inf))
    return Q


def _sensitivity(cfg, P, variables, nsteps, kind):
    a = _run(cfg, P, nsteps, kind)
    b = _run(cfg, _one_ulp(P, variables), nsteps, kind)
    scale = np.abs(a).reshape(cfg.nvar, -1).max(axis=1).reshape(-1, 1, 1, 1) + 1e-300
    return float((np.abs(a - b) / scale).max())


KINDS = ["orc"] + (["ref"] if have_ref() else [])


@pytest.mark.parametrize("kind", KINDS)
def test_reference_ideal_mhd_hlld_is_discontinuous_on_the_symmetric_blast(kind):
    cfg, P = problems.mhd_blastwave(24, 3, abi.EQMHD, abi.FLUX_RS_HLLD, strict_fp=1)
    s = _sensitivity(cfg, P, (abi.RO, abi.PG, abi.BX, abi.BY), 2, kind)
    assert s > 1e-7, "1 ulp in -> %g out: expected an O(1e-4) jump (branch flips)" % s


@pytest.mark.parametrize("kind", KINDS)
def test_reference_hll_and_glm_are_continuous_on_the_symmetric_blast(kind):
    cfg, P = problems.mhd_blastwave(24, 3, abi.EQMHD, abi.FLUX_RS_HLL, strict_fp=1)
    assert _sensitivity(cfg, P, (abi.RO, abi.PG, abi.BX, abi.BY), 2, kind) < 1e-13
    cfg, P = problems.mhd_blastwave(24, 3, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=1)
    assert _sensitivity(cfg, P, (abi.RO, abi.PG, abi.BX, abi.BY), 2, kind) < 1e-12


@pytest.mark.parametrize("eq", [abi.EQMHD, abi.EQGLM])
def test_reference_is_continuous_on_the_generic_blast(eq):
    cfg, P = problems.mhd_blast_generic([24, 24, 24], eq, abi.FLUX_RS_HLLD, strict_fp=1)
    s = _sensitivity(cfg, P, tuple(range(8)), 4, "orc")
    assert s < 1e-12, s
