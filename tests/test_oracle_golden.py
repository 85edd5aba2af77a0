"""CPU tests: the oracle (our restatement) against the golden fixtures generated from the
reference's own solver objects (tests/golden/make_golden.py).  Bit-exact."""
import os

import numpy as np
import pytest

import golden_cases as gc
from cpu_backends import CpuSim
from pion_amd import abi, driver

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def flux_kat():
    """flux_kat.npz and flux_kat_b.npz (Roe-MHD / linear MHD, added later) as one key -> array map"""
    d = {}
    for f in ("flux_kat.npz", "flux_kat_b.npz"):
        z = np.load(os.path.join(GOLD, f))
        d.update({k: z[k] for k in z.files})
    return d


@pytest.fixture(scope="module")
def cell_kat():
    return np.load(os.path.join(GOLD, "cell_kat.npz"))


@pytest.fixture(scope="module")
def steps():
    d = {}
    for f in ("steps.npz", "steps_b.npz"):
        z = np.load(os.path.join(GOLD, f))
        d.update({k: z[k] for k in z.files})
    return d


@pytest.mark.parametrize("case", gc.flux_cases() + gc.flux_cases_b(), ids=lambda c: gc.flux_key(*c))
def test_flux_kat(flux_kat, case):
    eq, sv, ntr, av = case
    key = gc.flux_key(*case)
    cfg = gc.flux_cfg(*case)
    L, R, aux = flux_kat[key + "_L"], flux_kat[key + "_R"], flux_kat[key + "_aux"]
    with CpuSim(cfg, "orc") as o:
        o.set_glm_speeds(gc.GLM_DT, cfg.dx, 0.25 / cfg.dx)
        for ax in range(3):
            F, _ = o.interface_flux(ax, L, R, aux, dt=gc.GLM_DT)
            assert np.array_equal(F, flux_kat[key + "_F%d" % ax], equal_nan=True), (key, ax)


@pytest.mark.parametrize("case", gc.cell_cases(), ids=lambda c: gc.cell_key(*c))
def test_cell_kat(cell_kat, case):
    key = gc.cell_key(*case)
    cfg = gc.cell_cfg(*case)
    P, dU = cell_kat[key + "_P"], cell_kat[key + "_dU"]
    with CpuSim(cfg, "orc") as o:
        o.set_glm_speeds(gc.GLM_DT, cfg.dx, 0.25 / cfg.dx)
        assert np.array_equal(o.cell_advance(P, dU, fv_dt=gc.GLM_DT), cell_kat[key + "_Pf"], equal_nan=True)
        assert np.array_equal(o.cell_timestep(P), cell_kat[key + "_dt"], equal_nan=True)


@pytest.mark.parametrize("name", gc.STEP_CASES + gc.STEP_CASES_B)
def test_whole_steps(steps, name):
    cfg, P = gc.step_case(name)
    with CpuSim(cfg, "orc") as o:
        gc.step_setup(name, o)
        sc = driver.SimControl(o, cfg)
        sc.init(P)
        assert np.array_equal(o.download(0), steps[name + "_bc"]), "boundary assignment"
        for it in range(gc.NSTEPS):
            dt = sc.calculate_timestep()
            assert dt == steps[name + "_dt"][it], (it, dt, steps[name + "_dt"][it])
            sc.advance_time()
        assert np.array_equal(o.download(0), steps[name + "_P"]), name


def test_fixtures_cover_all_flux_solvers(flux_kat):
    keys = {k.rsplit("_", 1)[0] for k in flux_kat}
    for c in gc.flux_cases() + gc.flux_cases_b():
        assert gc.flux_key(*c) in keys
