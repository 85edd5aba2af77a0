"""GPU cooling kernels and the Wind3D configuration against the oracle (strict: bit-exact)."""
import numpy as np
import pytest

from pion_amd import abi, cooling, driver, problems

pytestmark = pytest.mark.gpu


def _gpu(cfg):
    from pion_amd import lib
    return lib.GpuSim(cfg, 0)


def _cpu(cfg):
    from cpu_backends import CpuSim
    return CpuSim(cfg, "orc")


def _wind_cfg(strict):
    return problems.wind3d(16, strict_fp=strict)


def test_edot_and_time_update_strict():
    cfg, P, _, _ = _wind_cfg(1)
    T, tabs, sl = cooling.build_tables(cfg.min_temp, cfg.max_temp)
    rng = np.random.default_rng(3)
    n = 4000
    rho = 10.0 ** rng.uniform(-26, -20, n)
    Tk = 10.0 ** rng.uniform(3, 8.5, n)      # includes values outside the table
    Pin = np.zeros((n, cfg.nvar))
    Pin[:, abi.RO] = rho
    Pin[:, abi.PG] = rho * Tk / (0.609 * 1.672621898e-24 / 1.38064852e-16)
    Pin[:, abi.VX:abi.VZ + 1] = rng.normal(0, 1e6, (n, 3))
    with _gpu(cfg) as g, _cpu(cfg) as o:
        g.set_cooling_tables(T, tabs, sl)
        o.set_cooling_tables(T, tabs, sl)
        assert np.array_equal(g.cooling_edot(rho, Tk), o.cooling_edot(rho, Tk))
        for dt in (1.0e8, 1.0e10, 1.0e12):
            assert np.array_equal(g.cooling_update(Pin, dt), o.cooling_update(Pin, dt)), dt


@pytest.mark.parametrize("strict", [1, 0])
def test_wind3d_steps(strict):
    cfg, P, (idx, st), dtl = _wind_cfg(strict)
    T, tabs, sl = cooling.build_tables(cfg.min_temp, cfg.max_temp)
    with _gpu(cfg) as g, _cpu(cfg) as o:
        for s in (g, o):
            s.set_cooling_tables(T, tabs, sl)
            s.set_wind_cells(idx, st)
        sg, so = driver.SimControl(g, cfg), driver.SimControl(o, cfg)
        sg.first_step_dt_limit = so.first_step_dt_limit = dtl
        sg.init(P)
        so.init(P)
        for it in range(4):
            dg, do = sg.calculate_timestep(), so.calculate_timestep()
            if strict:
                assert dg == do, (it, dg, do)
            else:
                assert abs(dg - do) <= 1e-10 * do
            so.dt = sg.dt
            sg.advance_time()
            so.advance_time()
            a, b = g.download(0), o.download(0)
            if strict:
                assert np.array_equal(a, b), (it, (a != b).sum())
            else:
                scale = np.abs(b).reshape(cfg.nvar, -1).max(axis=1).reshape(-1, 1, 1, 1)
                assert np.max(np.abs(a - b) / scale) <= 1e-10


def test_cooling_tables_size_is_checked():
    """k_cooling_dE holds the tables in LDS (11 x 256 doubles): larger tables are refused, not truncated"""
    from pion_amd import lib
    cfg, P, (idx, st), dtl = problems.wind3d(8, strict_fp=1)
    with lib.GpuSim(cfg, 0) as g:
        n = 300
        T = np.logspace(1, 9, n)
        with pytest.raises(Exception):
            g.set_cooling_tables(T, np.ones((5, n)), np.zeros((5, n)))


def test_table_interval_from_the_log_guess_equals_the_bisection(monkeypatch):
    """Edot's table interval: the reference bisects; on the log-spaced grid the device starts from a single-precision
    logarithm and corrects against the table (same index by construction).  Bit-identical to the bisecting path
    (PION_COOL_BISECT=1) and to the oracle at table nodes, one ulp either side of them, outside the table, and for a
    table that is NOT log-spaced (where the device falls back to bisecting)."""
    cfg, P, _, _ = _wind_cfg(1)
    T, tabs, sl = cooling.build_tables(cfg.min_temp, cfg.max_temp)
    rng = np.random.default_rng(11)
    n = 6000
    rho = 10.0 ** rng.uniform(-26, -20, n)
    Tk = 10.0 ** rng.uniform(2.5, 9.0, n)
    k = rng.integers(0, T.size, 600)
    Tk[:200] = T[k[:200]]
    Tk[200:400] = np.nextafter(T[k[200:400]], 0.0)
    Tk[400:600] = np.nextafter(T[k[400:600]], np.inf)
    Tk[600:610] = [T[0], T[-1], np.nextafter(T[0], 0), np.nextafter(T[-1], np.inf), 1e-30, 1e30, 1e300, 1e-300, 5e-324, T[1]]
    out = {}
    for mode in ("log", "bisect"):
        if mode == "bisect":
            monkeypatch.setenv("PION_COOL_BISECT", "1")
        else:
            monkeypatch.delenv("PION_COOL_BISECT", raising=False)
        with _gpu(cfg) as g:
            g.set_cooling_tables(T, tabs, sl)
            out[mode] = g.cooling_edot(rho, Tk)
    monkeypatch.delenv("PION_COOL_BISECT", raising=False)
    with _cpu(cfg) as o:
        o.set_cooling_tables(T, tabs, sl)
        want = o.cooling_edot(rho, Tk)
    assert np.array_equal(out["log"], out["bisect"])
    assert np.array_equal(out["log"], want)
    # a linearly spaced table: not log-spaced -> bisection on the device; still the oracle's answer
    Tl = np.linspace(5.0e3, 1.0e8, 200)
    tl = np.abs(rng.normal(1e-23, 3e-24, (5, 200)))
    sll = np.zeros((5, 200))
    sll[:, :-1] = np.diff(tl, axis=1) / np.diff(Tl)
    with _gpu(cfg) as g, _cpu(cfg) as o:
        g.set_cooling_tables(Tl, tl, sll)
        o.set_cooling_tables(Tl, tl, sll)
        assert np.array_equal(g.cooling_edot(rho, Tk), o.cooling_edot(rho, Tk))
