"""More of the reference's SHIPPED uniform-grid test problems (tests/golden/endstate_s.npz, made by
tests/golden/make_golden.py f from oracle/_ref, i.e. the reference's own solver objects): FieldLoop100 (run to its
shipped finish time) / vz / Static, the StarBench contact-discontinuity advection, the Liska-Wendroff implosion, the
oblique (Quirk-unstable) shocks at Mach 25 / 40 with the three solver / viscosity pairs of run_ObliqueShockTest.sh, and
the axisymmetric blast waves (Euler: the whole NR016 run; glm-mhd with the Roe solver).  Parameter files and IC
functions are cited in tests/golden_cases.py::end_case_s.

CPU: the oracle reproduces every dt and the end state bit for bit.  GPU: so does the strict build through the C-ABI;
the fast (benchmarked) build is held to the SURVEY 8(d) gates (L1 / L2 <= 1e-10 x refvec, conserved totals 1e-10)."""
import os

import numpy as np
import pytest

import golden_cases as gc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "endstate_s.npz")


@pytest.fixture(scope="module")
def gold():
    z = np.load(GOLD)
    return {k: z[k] for k in z.files}


def test_every_case_has_a_fixture(gold):
    for name in gc.END_CASES_S:
        assert name + "_P" in gold, name
    # the cases that reach the shipped finish time
    assert float(gold["fieldloop100_t"]) == 2.0
    assert float(gold["bwaxi2d_halfplane_nr016_t"]) == 1.58e12


@pytest.mark.parametrize("name", gc.END_CASES_S)
def test_oracle_reproduces_reference_end_state(gold, name):
    from cpu_backends import CpuSim
    cfg, P, tf, nmax = gc.end_case_s(name)
    with CpuSim(cfg, "orc") as o:
        n, t, dts = gc.end_run(o, cfg, P, tf, nmax)
        A = o.download(0)
    assert n == int(gold[name + "_n"]) and t == float(gold[name + "_t"])
    assert np.array_equal(dts, gold[name + "_dt"])
    assert np.array_equal(A, gold[name + "_P"]), name
    tot, _ = gc.conserved_totals(cfg, A)
    assert np.array_equal(tot, gold[name + "_tot"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", gc.END_CASES_S)
def test_gpu_strict_reproduces_reference_end_state(gold, name):
    from pion_amd import lib
    cfg, P, tf, nmax = gc.end_case_s(name, strict_fp=1)
    with lib.GpuSim(cfg, 0) as g:
        n, t, dts = gc.end_run(g, cfg, P, tf, nmax)
        A = g.download(0)
    assert n == int(gold[name + "_n"]) and t == float(gold[name + "_t"])
    assert np.array_equal(dts, gold[name + "_dt"])
    assert np.array_equal(A, gold[name + "_P"]), name


# per case: (L1 and L2 / refvec, conserved totals relative to the sum of magnitudes); the SURVEY 8(d) gate is 1e-10
FAST_TOL = {name: (1e-10, 1e-10) for name in gc.END_CASES_S}
# Roe-CV + FKJ98 on the nearly grid-aligned Mach-25 shock is the Quirk-UNSTABLE pair this shipped test exists to show
# (test_problems/ObliqueShock/README.txt): the reference itself turns a 1-ulp change of its input into L2 = 2.5e-10
# after these 250 steps (test_reference_amplifies_rounding_on_the_quirk_unstable_shock below); the H-correction
# and FVS runs of the same shock stay at the 1e-10 gate
FAST_TOL["oblique_m25_roe_fkj"] = (1e-9, 1e-10)


def _one_ulp_sensitivity(name):
    from cpu_backends import CpuSim
    cfg, P, tf, nmax = gc.end_case_s(name)
    rng = np.random.default_rng(1)
    Q = P.copy()
    for v in (0, 1, 2):
        up = rng.integers(0, 2, Q[v].shape) > 0
        Q[v] = np.where(up, np.nextafter(Q[v], np.inf), np.nextafter(Q[v], -np.inf))
    out = []
    for X in (P, Q):
        with CpuSim(cfg, "orc") as o:
            gc.end_run(o, cfg, X, tf, nmax)
            out.append(o.download(0).copy())
    return gc.diff_norms(cfg, out[0], out[1])


def test_reference_amplifies_rounding_on_the_quirk_unstable_shock():
    """(CPU, the oracle = the reference bit for bit) 1 ulp in -> L2 > 1e-10 out for Roe-CV + FKJ98, two orders less
    with the H-correction: why the fast build's gate for that one case is 1e-9"""
    _, l2, _ = _one_ulp_sensitivity("oblique_m25_roe_fkj")
    assert l2.max() > 1e-10, l2
    _, l2h, _ = _one_ulp_sensitivity("oblique_m25_roe_hcorr")
    assert l2h.max() < 2e-11, l2h


def _totals_rel(cfg, tot, gold_tot, mag):
    """|difference| of each conserved total over its scale: the sum of magnitudes of the component -- for a vector
    (momentum, field) of its largest component, so that a component that vanishes by symmetry (v_theta, B_theta, B_R of
    the axisymmetric blast: totals and magnitudes at rounding level) is measured against the vector it belongs to"""
    den = mag.copy()
    den[1:4] = mag[1:4].max()
    if len(mag) >= 8:   # (mass, momentum x 3, field x 3, energy)
        den[4:7] = mag[4:7].max()
    return np.abs(tot - gold_tot) / (den + 1e-300)


@pytest.mark.gpu
@pytest.mark.parametrize("name", gc.END_CASES_S)
def test_gpu_fast_build_end_state_norms(gold, name):
    from pion_amd import lib
    cfg, P, tf, nmax = gc.end_case_s(name, strict_fp=0)
    with lib.GpuSim(cfg, 0) as g:
        n, t, dts = gc.end_run(g, cfg, P, tf, nmax)
        A = g.download(0)
    assert np.isfinite(A).all()
    assert n == int(gold[name + "_n"]), (n, int(gold[name + "_n"]))
    assert abs(t - float(gold[name + "_t"])) <= 1e-9 * abs(t)
    tol_norm, tol_tot = FAST_TOL[name]
    tot, mag = gc.conserved_totals(cfg, A)
    rel = _totals_rel(cfg, tot, gold[name + "_tot"], mag)
    assert rel.max() <= tol_tot, rel
    l1, l2, mx = gc.diff_norms(cfg, A, gold[name + "_P"])
    assert l1.max() <= tol_norm and l2.max() <= tol_norm, (name, l1, l2, mx)
