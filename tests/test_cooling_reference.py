"""The cooling path (SURVEY 8a rows A23 TimeUpdateMP, A24 Edot + tables, A26 timescales) against outputs of
the REFERENCE's own mp_only_cooling object.

Fixtures (tests/golden/make_golden.py d): microphysics/mp_only_cooling.cpp compiled as it lies into
oracle/_ref; the three GSL-spline-backed rate curves of its base classes are test doubles fed with the
product's curves (oracle/ref_cooling.cpp) -- so what is pinned to the reference is everything DOWNSTREAM of
the curves: the table grid / layout / slopes of gen_mpoc_lookup_tables, the bisection + interpolation +
rate formula of Edot, TimeUpdateMP with its clamps and the Euler shortcut, timescales, and whole time steps /
a 60-step run with the reference object as the global MP.  What stays PARITY UNPINNED: the values of the three
spline curves themselves (tests/test_cooling.py checks them against scipy's natural cubic spline).

CPU part: the product's table builder and the oracle are bit-identical to the fixtures.
GPU part (-m gpu): strict build bit-identical; fast (benchmarked) build <= 1e-10 relative."""
import os

import numpy as np
import pytest

import golden_cases as gc
from pion_amd import abi, cooling, driver

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RANGES = list(range(len(gc.COOL_RANGES)))


@pytest.fixture(scope="module")
def kat():
    z = np.load(os.path.join(GOLD, "cooling_kat.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def steps_c():
    z = np.load(os.path.join(GOLD, "steps_c.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def end_c():
    z = np.load(os.path.join(GOLD, "endstate_c.npz"))
    return {k: z[k] for k in z.files}


def _orc(cfg):
    from cpu_backends import CpuSim
    return CpuSim(cfg, "orc")


def _gpu(cfg):
    from pion_amd import lib
    return lib.GpuSim(cfg, 0)


def _tables(cfg):
    return cooling.build_tables(cfg.min_temp, cfg.max_temp)


# ------------------------------------------------------------------------------------------- CPU
@pytest.mark.parametrize("k", RANGES)
def test_table_builder_reproduces_reference_tables(kat, k):
    """pion_host_build_cooling_tables == mp_only_cooling::gen_mpoc_lookup_tables, bit for bit: the 200-point
    temperature grid, the five tables and the five slope tables (last slope 0)"""
    tmin, tmax = gc.COOL_RANGES[k]
    T, tabs, sl = cooling.build_tables(tmin, tmax)
    key = "r%d_" % k
    assert np.array_equal(T, kat[key + "T"])
    assert np.array_equal(tabs, kat[key + "tabs"])
    assert np.array_equal(sl, kat[key + "slopes"])
    # MinT_allowed / MaxT_allowed as the reference's constructor overrides them (mp_only_cooling.cpp:140-146)
    lim = kat[key + "limits"]
    want_min = 1.0 if (tmin < 1.0 or tmin > 1.0e6) else tmin
    want_max = 1.0e8 if (tmax < 1.0e2 or tmax > 3.0e10) else tmax
    assert lim[0] == want_min and lim[1] == want_max
    assert lim[2] == gc.ODE_MU_TOT_OVER_KB


def _check_cooling_kats(sim, kat, k, exact, tol=0.0):
    key = "r%d_" % k
    e = sim.cooling_edot(kat[key + "edot_rho"], kat[key + "edot_T"])
    want = kat[key + "edot"]
    P = kat[key + "P"]
    tc = sim.cooling_timescale(P)
    if exact:
        assert np.array_equal(e, want)
        assert np.array_equal(tc, kat[key + "tcool"])
    else:
        # Edot is a difference of heating and cooling terms: relative to the larger of them
        scale = np.abs(want) + 1e-12 * np.abs(want).max()
        assert np.max(np.abs(e - want) / scale) <= 1e-9
        assert np.allclose(tc, kat[key + "tcool"], rtol=1e-9, atol=0.0)
    for j, dt in enumerate(gc.COOL_DTS):
        ok = kat[key + "upd%d_ok" % j]
        out = sim.cooling_update(P[ok], dt)
        wantP = kat[key + "upd%d_P" % j]
        if exact:
            assert np.array_equal(out, wantP), (k, dt)
        else:
            assert np.allclose(out, wantP, rtol=tol, atol=0.0), (k, dt)


@pytest.mark.parametrize("k", RANGES)
def test_oracle_edot_update_timescales_match_reference(kat, k):
    cfg = gc.cool_cfg(k)
    with _orc(cfg) as o:
        o.set_cooling_tables(*_tables(cfg))
        _check_cooling_kats(o, kat, k, exact=True)


def test_cooling_kat_covers_the_branches(kat):
    """the fixture exercises: both final temperature clamps, the Euler shortcut and the Cash-Karp path, states
    below 1.1 MinT_allowed in timescales, temperatures outside the table"""
    for k in RANGES:
        key = "r%d_" % k
        lim = kat[key + "limits"]
        Tf_all = np.concatenate([kat[key + "upd%d_Tf" % j] for j in range(len(gc.COOL_DTS))])
        assert (Tf_all == lim[0]).any(), "lower clamp"
        assert (Tf_all == lim[1]).any(), "upper clamp"
        assert (kat[key + "tcool"] == 1.0e99).any() and (kat[key + "tcool"] < 1.0e99).any()
        T = kat[key + "edot_T"]
        assert (T < kat[key + "T"][0]).any() and (T > kat[key + "T"][-1]).any()
        # smallest dt: states for which |Edot| dt / E < 1e-6, the Euler shortcut of integrator.cpp:325-339
        # (E / |Edot| >= t_cool by the definition of timescales), and states that take the Cash-Karp path
        tc = kat[key + "tcool"]
        assert (gc.COOL_DTS[0] / tc[tc < 1.0e99] < 1e-6).any()
        assert (gc.COOL_DTS[-1] / tc[tc < 1.0e99] > 1.0).any()


@pytest.mark.parametrize("name", gc.STEP_CASES_C)
def test_oracle_whole_steps_with_reference_cooling(steps_c, name):
    cfg, P = gc.step_case_c(name)
    with _orc(cfg) as o:
        o.set_cooling_tables(*_tables(cfg))
        sc = driver.SimControl(o, cfg)
        sc.init(P)
        for it in range(gc.NSTEPS):
            assert sc.calculate_timestep() == steps_c[name + "_dt"][it]
            sc.advance_time()
        assert np.array_equal(o.download(0), steps_c[name + "_P"])


def _end_run(sim, name, strict_fp=1):
    cfg, P, tf, nmax = gc.end_case_c(name, strict_fp=strict_fp)
    if cfg.cooling:
        sim.set_cooling_tables(*_tables(cfg))
    return gc.end_run(sim, cfg, P, tf, nmax)


@pytest.mark.parametrize("name", gc.END_CASES_C)
def test_oracle_reproduces_reference_cooling_end_state(end_c, name):
    cfg, P, tf, nmax = gc.end_case_c(name)
    with _orc(cfg) as o:
        n, t, dts = _end_run(o, name)
        A = o.download(0)
    assert n == int(end_c[name + "_n"]) and t == float(end_c[name + "_t"])
    assert np.array_equal(dts, end_c[name + "_dt"])
    assert np.array_equal(A, end_c[name + "_P"])
    if cfg.cooling:
        # the cooling-time limit binds in this run (dt < the CFL step) and the gas has lost energy
        tot0, _ = gc.conserved_totals(cfg, P)
        assert end_c[name + "_tot"][-1] < 0.9 * tot0[-1]


def test_fixture_is_what_the_reference_gives_now(kat):
    """where oracle/_ref is built (the build container): the committed fixture equals the live reference object"""
    from cpu_backends import have_ref, RefCooling
    if not have_ref():
        pytest.skip("oracle/_ref is only built where /root/reference exists")
    tmin, tmax = gc.COOL_RANGES[0]
    with RefCooling(tmin, tmax, gc.COOL_GAMMA, gc.COOL_NVAR, 1) as r:
        T, tabs, sl = r.tables()
        assert np.array_equal(T, kat["r0_T"]) and np.array_equal(tabs, kat["r0_tabs"])
        assert np.array_equal(r.edot(kat["r0_edot_rho"], kat["r0_edot_T"]), kat["r0_edot"])
        assert np.array_equal(r.timescale(kat["r0_P"]), kat["r0_tcool"])
        out, Tf = r.update(kat["r0_P"][kat["r0_upd2_ok"]], gc.COOL_DTS[2])
        assert np.array_equal(out, kat["r0_upd2_P"])


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("k", RANGES)
@pytest.mark.parametrize("strict", [1, 0])
def test_gpu_edot_update_timescales_match_reference(kat, k, strict):
    cfg = gc.cool_cfg(k, strict_fp=strict)
    with _gpu(cfg) as g:
        g.set_cooling_tables(*_tables(cfg))
        _check_cooling_kats(g, kat, k, exact=bool(strict), tol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("name", gc.STEP_CASES_C)
def test_gpu_whole_steps_with_reference_cooling(steps_c, name):
    cfg, P = gc.step_case_c(name, strict_fp=1)
    with _gpu(cfg) as g:
        g.set_cooling_tables(*_tables(cfg))
        sc = driver.SimControl(g, cfg)
        sc.init(P)
        for it in range(gc.NSTEPS):
            assert sc.calculate_timestep() == steps_c[name + "_dt"][it]
            sc.advance_time()
        assert np.array_equal(g.download(0), steps_c[name + "_P"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", gc.END_CASES_C)
@pytest.mark.parametrize("strict", [1, 0])
def test_gpu_reproduces_reference_cooling_end_state(end_c, name, strict):
    cfg, P, tf, nmax = gc.end_case_c(name, strict_fp=strict)
    with _gpu(cfg) as g:
        n, t, dts = _end_run(g, name, strict_fp=strict)
        A = g.download(0)
    assert n == int(end_c[name + "_n"])
    if strict:
        assert t == float(end_c[name + "_t"])
        assert np.array_equal(dts, end_c[name + "_dt"])
        assert np.array_equal(A, end_c[name + "_P"])
    else:
        assert np.allclose(dts, end_c[name + "_dt"], rtol=1e-9, atol=0.0)
        l1, l2, mx = gc.diff_norms(cfg, A, end_c[name + "_P"])
        assert l1.max() <= 1e-10 and l2.max() <= 1e-10, (l1, l2, mx)
        tot, mag = gc.conserved_totals(cfg, A)
        assert (np.abs(tot - end_c[name + "_tot"]) / (mag + 1e-300)).max() <= 1e-10
