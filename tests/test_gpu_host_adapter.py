"""The C++ host adapter (pion_amd/host/sim_control_gpu.cpp: Time_Int / calculate_timestep /
advance_time mirror) must give exactly what the Python driver gives."""
import ctypes as C
import os

import numpy as np
import pytest

from pion_amd import abi, driver, problems

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_time_int_equals_python_driver():
    from pion_amd import lib
    abi.share_torch_hip_runtime()
    host = C.CDLL(os.path.join(ROOT, "pion_amd", "host", "libpion_host.so"))
    dp = C.POINTER(C.c_double)
    host.pion_host_sim_create.argtypes = [C.POINTER(abi.PionGpuConfig), C.c_int, C.POINTER(C.c_void_p)]
    host.pion_host_sim_init.argtypes = [C.c_void_p, dp, C.c_double, C.c_double, C.c_double]
    host.pion_host_sim_time_int.argtypes = [C.c_void_p, C.c_int, dp, dp]
    host.pion_host_sim_download.argtypes = [C.c_void_p, C.c_int, dp]
    host.pion_host_sim_destroy.argtypes = [C.c_void_p]
    host.pion_host_sim_destroy.restype = None
    # (3-D periodic GLM-MHD, 3-D Euler with physical faces, and two 2-D grids: Cartesian and cylindrical (z,R))
    for cfg, P in (problems.mhd_blastwave(20, 3, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=1),
                   problems.hd_blast_octant(20, 3, solver=abi.FLUX_RSroe, strict_fp=1, nzones=3.0),
                   problems.mhd_blastwave(40, 2, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=1),
                   problems.blast_axi2d(72, abi.EQGLM, abi.FLUX_RS_HLLD, ntracer=1, strict_fp=1)):
        s = C.c_void_p()
        assert host.pion_host_sim_create(C.byref(cfg), 0, C.byref(s)) == 0
        Pc = np.ascontiguousarray(P).reshape(-1)
        assert host.pion_host_sim_init(s, Pc.ctypes.data_as(dp), 0.0, 1e300, -1.0) == 0
        t, ldt = C.c_double(), C.c_double()
        assert host.pion_host_sim_time_int(s, 4, C.byref(t), C.byref(ldt)) == 4
        out = np.empty_like(Pc)
        assert host.pion_host_sim_download(s, 0, out.ctypes.data_as(dp)) == 0
        host.pion_host_sim_destroy(s)
        with lib.GpuSim(cfg, 0) as g:
            sc = driver.SimControl(g, cfg)
            sc.init(P)
            sc.time_int(4)
            assert sc.simtime == t.value and sc.last_dt == ldt.value
            assert np.array_equal(g.download(0).reshape(-1), out)


def test_second_init_on_the_same_object_does_not_see_the_old_time_step():
    """advance_time() ends by requesting the next step's minima; a second Init (new problem, restart) must not
    consume that request: the first dt after it comes from the state uploaded last."""
    from pion_amd import host_rccl
    cfg, P = problems.mhd_blastwave(20, 3, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=1)
    P2 = P.copy()
    P2[abi.PG] *= 7.0          # a different problem: shorter time step
    P2[abi.VX] += 0.3
    with host_rccl.HostSim(cfg, 0) as fresh:
        fresh.init(P2)
        nf, tf, lf = fresh.time_int(3)
        want = fresh.download(0)
    with host_rccl.HostSim(cfg, 0) as s:
        s.init(P)
        s.time_int(2)
        s.init(P2)                       # same object, new state, time reset
        n, t, l = s.time_int(3)
        got = s.download(0)
    assert (n, t, l) == (nf, tf, lf)
    assert np.array_equal(got, want)


def test_restart_through_the_host_driver_is_bit_identical():
    """init, 2 steps, download, Init again from the downloaded state with the time bookkeeping of a snapshot,
    3 more steps == 5 uninterrupted steps"""
    from pion_amd import host_rccl
    cfg, P = problems.hd_blast_octant(20, 3, solver=abi.FLUX_RSroe, strict_fp=1, nzones=3.0)
    with host_rccl.HostSim(cfg, 0) as a:
        a.init(P)
        a.time_int(5)
        want = a.download(0)
    with host_rccl.HostSim(cfg, 0) as b:
        b.init(P)
        n, t, l = b.time_int(2)
        mid = b.download(0)
        b.init(mid, simtime=t, timestep=2, last_dt=l)
        b.time_int(3)
        assert np.array_equal(b.download(0), want)
