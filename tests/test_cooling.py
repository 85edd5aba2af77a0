"""Cooling source term (mp_only_cooling, EP.cooling=8): host table builder, oracle ODE
integrator against the reference's own Integrator_Base (golden ode_kat.npz)."""
import os

import numpy as np
import pytest

import golden_cases as gc
from cpu_backends import CpuSim
from pion_amd import abi, cooling, driver, problems

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ode_cfg(strict_fp=1):
    return abi.make_config(3, [4, 4, 4], abi.EQEUL, abi.FLUX_FVS, ntracer=0, xmax=(1, 1, 1),
                           gamma=gc.ODE_GAMMA, cooling=abi.COOL_WSS09_CIE_LINE_HEAT_COOL,
                           min_temp=gc.ODE_TMIN, max_temp=gc.ODE_TMAX, strict_fp=strict_fp)


def test_spline_matches_scipy_natural_cubic():
    """PARITY UNPINNED against the reference (GSL is absent): the natural cubic spline of the
    table builder is cross-checked against scipy's independent implementation instead."""
    from scipy.interpolate import CubicSpline
    import re
    hdr = open(os.path.join(os.path.dirname(GOLD), "..", "pion_amd", "host", "cooling_data.h")).read()

    def arr(name):
        m = re.search(r"%s\[\d+\] = \{(.*?)\};" % name, hdr, re.S)
        return np.array([float(x) for x in m.group(1).replace("\n", " ").split(",")])
    lt, ll = arr("WSS09_logT"), arr("WSS09_logL")
    cs = CubicSpline(lt, ll, bc_type="natural")
    for T in 10.0 ** np.linspace(2.01, 8.97, 400):
        want = np.exp(2.3025850929940459 * cs(np.log10(T)))
        got = cooling.cooling_rate_wss09(T)
        assert abs(got - want) <= 1e-12 * want
    # extrapolation branches (cooling_SD93_cie.cpp:684-693)
    assert abs(cooling.cooling_rate_wss09(10.0) - 10 ** (ll[0] + 8.0 * (1.0 - lt[0]))) < 1e-12 * cooling.cooling_rate_wss09(10.0)
    hT = np.exp(np.log(10.0) * (1.0 + 0.2 * np.arange(31)))
    ca = arr("H94_caseB") / np.sqrt(hT)
    csa = CubicSpline(hT, ca, bc_type="natural")
    for T in 10.0 ** np.linspace(1.01, 6.99, 300):
        assert abs(cooling.hii_rrr(T) - csa(T)) <= 1e-11 * abs(csa(T))
    assert 2.4e-13 < cooling.hii_rrr(1.0e4) < 2.7e-13   # case-B alpha at 1e4 K


def test_table_layout_and_slopes():
    T, tabs, sl = cooling.build_tables(5.0e3, 1.0e8, 200)
    assert abs(T[0] - 5.0e3) < 1e-8 and abs(T[-1] - 1.0e8) < 1e-4
    assert np.all(np.diff(T) > 0) and np.all(tabs[[0, 1, 2, 4]] > 0)
    for k in range(5):
        assert np.allclose(sl[k, :-1], np.diff(tabs[k]) / np.diff(T), rtol=1e-15)
        assert sl[k, -1] == 0.0
    assert np.allclose(tabs[2], 6.72e-28 * np.sqrt(T), rtol=1e-15)


def test_oracle_integrator_matches_reference_integrator():
    """mp_only_cooling::TimeUpdateMP in the oracle vs the reference's Int_Adaptive_RKCK on the same
    piecewise-linear rate (fixture from oracle/_ref).  The rate evaluations differ by rounding of the
    E<->T mapping, so compare to 1e-9 and only where the reference reported success."""
    d = np.load(os.path.join(GOLD, "ode_kat.npz"))
    cfg = _ode_cfg()
    T, tabs, sl = gc.ode_cooling_tables()
    # reference reported success AND reached the end time within its 25 adaptive steps
    ok = (d["errs"] == 0) & (d["tout"] >= d["dt"] * (1 - 1e-12))
    assert ok.sum() > 200
    with CpuSim(cfg, "orc") as o:
        o.set_cooling_tables(T, tabs, sl)
        for i in np.flatnonzero(ok)[:250]:
            P = np.zeros((1, 5))
            P[0, abi.RO] = gc.ODE_RHO
            P[0, abi.PG] = d["E0"][i] * (gc.ODE_GAMMA - 1.0)
            out = o.cooling_update(P, float(d["dt"][i]))
            E = out[0, abi.PG] / (gc.ODE_GAMMA - 1.0)
            Tf = out[0, abi.PG] * gc.ODE_MU_TOT_OVER_KB / gc.ODE_RHO
            if Tf <= gc.ODE_TMIN * (1 + 1e-12):
                continue  # clamped to MinT_allowed by TimeUpdateMP, the bare integrator is not
            assert abs(E - d["Eout"][i]) <= 1e-9 * d["E0"][i], (i, E, d["Eout"][i])


def test_wind3d_runs_on_oracle_and_cools():
    cfg, P, (idx, st), dtl = problems.wind3d(12, strict_fp=1)
    T, tabs, sl = cooling.build_tables(cfg.min_temp, cfg.max_temp)
    with CpuSim(cfg, "orc") as o:
        o.set_cooling_tables(T, tabs, sl)
        o.set_wind_cells(idx, st)
        sc = driver.SimControl(o, cfg)
        sc.first_step_dt_limit = dtl
        sc.init(P)
        assert sc.calculate_timestep() == dtl         # first-step wind limiter active
        sc.advance_time()
        sc.time_int(3)
        A = o.download(0)
        assert np.isfinite(A).all()
        fl = o.flags().reshape(-1)
        assert ((fl[idx] & abi.CELL_ISBD) != 0).all() and ((fl[idx] & abi.CELL_ISDOMAIN) == 0).all()
        # on-grid wind cells keep their fixed state (ghost ones are refilled by the reflecting BC)
        on = (fl[idx] & abi.CELL_ISGD) != 0
        for v in range(cfg.nvar):
            assert np.array_equal(A[v].reshape(-1)[idx][on], st[on, v])
