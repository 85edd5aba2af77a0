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
