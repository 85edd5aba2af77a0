"""The shock tubes the reference ships as initial conditions (source/ics/shock_tube.cpp:473-815 =
dataIO/dataio_text.cpp:734-1080): Toro 1-5 (Euler, hybrid Riemann solver, FKJ98 viscosity), Brio & Wu and Falle's
FS / SS / FR / SR / OFS, Ryu & Jones 1a-5b (ideal MHD, HLLD, no viscosity); 1-D, 200 cells, outflow, CFL 0.7, run to
their finish times (76-823 steps).  tests/golden/shocktubes.npz holds every dt and the end state from the REFERENCE's
own solver objects (tests/golden/make_golden.py e).

CPU: the oracle reproduces them bit for bit.  GPU: the strict build does (hybrid solver: device exp / log / pow,
<= 1e-9), the fast build is held to L1, L2 <= 1e-10 x refvec."""
import os

import numpy as np
import pytest

import golden_cases as gc
from pion_amd import abi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shocktubes.npz")


@pytest.fixture(scope="module")
def gold():
    z = np.load(GOLD)
    return {k: z[k] for k in z.files}


def test_every_shipped_shock_tube_has_a_fixture(gold):
    for name in gc.SHOCK_TUBES:
        assert name + "_P" in gold, name


@pytest.mark.parametrize("name", gc.SHOCK_TUBES)
def test_oracle_reproduces_reference_shock_tube(gold, name):
    from cpu_backends import CpuSim
    cfg, P, tf = gc.shock_tube_case(name)
    with CpuSim(cfg, "orc") as o:
        n, t, dts = gc.end_run(o, cfg, P, tf, 100000)
        A = o.download(0)
    assert n == int(gold[name + "_n"]) and t == float(gold[name + "_t"])
    assert np.array_equal(dts, gold[name + "_dt"])
    assert np.array_equal(A, gold[name + "_P"]), name
    # something happened: the end state is not the initial one
    assert not np.array_equal(A, P)


@pytest.mark.gpu
@pytest.mark.parametrize("name", gc.SHOCK_TUBES)
def test_gpu_strict_reproduces_reference_shock_tube(gold, name):
    from pion_amd import lib
    cfg, P, tf = gc.shock_tube_case(name, strict_fp=1)
    with lib.GpuSim(cfg, 0) as g:
        n, t, dts = gc.end_run(g, cfg, P, tf, 100000)
        A = g.download(0)
    if cfg.solver == abi.FLUX_RShybrid:
        # exp / log / pow of the device maths library differ from glibc in the last bits
        assert abs(n - int(gold[name + "_n"])) <= 1
        if n == int(gold[name + "_n"]):
            assert np.allclose(dts, gold[name + "_dt"], rtol=1e-9, atol=0.0)
            l1, l2, mx = gc.diff_norms(cfg, A, gold[name + "_P"])
            assert l1.max() <= 1e-9 and l2.max() <= 1e-9, (l1, l2, mx)
    else:
        assert n == int(gold[name + "_n"]) and t == float(gold[name + "_t"])
        assert np.array_equal(dts, gold[name + "_dt"])
        assert np.array_equal(A, gold[name + "_P"]), name


# Gate of the fast build per case: the SURVEY 8(d) 1e-10 x refvec, except where the REFERENCE ALGORITHM itself is
# that ill-conditioned: Ryu & Jones 4a (a switch-on fast shock: B_y = 0 exactly on one side, B_z = 0 on both)
# answers a 1-ulp change of its INPUT with 7e-10 (L1) / 2e-9 (L2) / 2e-8 (max) x refvec in the end state --
# test_reference_amplifies_one_ulp_in_rj4a below measures that on the oracle -- so no build-independent answer
# exists below that level and the gate there is 25 x the measured sensitivity.
FAST_GATE = {"rj4a": 5e-8}


def _one_ulp_sensitivity(name):
    from cpu_backends import CpuSim
    cfg, P, tf = gc.shock_tube_case(name)
    outs = []
    for pert in (0, 1):
        Q = P.copy()
        if pert:
            Q = np.nextafter(Q, np.inf)
            Q[P == 0] = 0.0
        with CpuSim(cfg, "orc") as o:
            gc.end_run(o, cfg, Q, tf, 100000)
            outs.append(o.download(0))
    return gc.diff_norms(cfg, outs[0], outs[1])


def test_reference_amplifies_one_ulp_in_rj4a():
    """conditioning of the shipped problems under the reference's own arithmetic (the oracle is bit-identical to
    it on these runs): one ulp on the initial state moves RJ 4a's end state by > 1e-10 x refvec, and a
    well-conditioned neighbour (RJ 4b) by < 1e-12"""
    l1, l2, mx = _one_ulp_sensitivity("rj4a")
    assert l2.max() > 1e-10 and l2.max() < 2e-9 * 5
    l1, l2, mx = _one_ulp_sensitivity("rj4b")
    assert l2.max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("name", gc.SHOCK_TUBES)
def test_gpu_fast_build_shock_tube_norms(gold, name):
    from pion_amd import lib
    cfg, P, tf = gc.shock_tube_case(name, strict_fp=0)
    with lib.GpuSim(cfg, 0) as g:
        n, t, dts = gc.end_run(g, cfg, P, tf, 100000)
        A = g.download(0)
    assert np.isfinite(A).all()
    assert n == int(gold[name + "_n"]), (n, int(gold[name + "_n"]))
    assert abs(t - float(gold[name + "_t"])) <= 1e-12 * abs(t)
    l1, l2, mx = gc.diff_norms(cfg, A, gold[name + "_P"])
    gate = FAST_GATE.get(name, 1e-10)
    assert l1.max() <= gate and l2.max() <= gate, (name, l1, l2, mx)
