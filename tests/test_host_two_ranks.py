"""The C++ host time loop with TWO RANKS on the CPU (world_size 2, no GPU): pion_host::sim_control_gpu + the split
stages + request_min / allreduce_min + pion_host::slab_comm_shm -- the code `bench.py --gpus N` executes, with the
transport's wire swapped from RCCL to host shared memory -- run as two processes whose backend table
(pion_amd/host/pion_backend.h) is bound to the oracle (tests/native/orc_backend.cpp; test infrastructure).  Must
reproduce the single-domain oracle run bit for bit: GLM-MHD periodic (world = 2: both neighbours are the same peer),
HD octant (physical z faces on the end ranks), Wind3D (cooling + wind cells + first-step limit).
It replaces comm_mpi.cpp:287-425 and sim_control_MPI.cpp:482-583 for this path.  (tests/test_slab_gloo.py covers the
Python driver over gloo; tests/test_gpu_host_two_ranks.py the same C++ code on the GPU.)"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")


def _orc_backend():
    so = os.path.join(NATIVE, "liborc_backend.so")
    subprocess.check_call(["make", "-s", "-C", NATIVE])
    lib = C.CDLL(so)
    lib.pion_backend_oracle.restype = C.c_void_p
    lib.pion_backend_oracle_handle.restype = C.c_void_p
    lib.pion_backend_oracle_handle.argtypes = [C.c_void_p]
    return lib


def _case(case):
    from pion_amd import abi, problems
    if case == "wind3d":
        cfg, P, _, _ = problems.wind3d(12, strict_fp=1)
        return cfg, P
    if case == "glm_periodic":
        return problems.mhd_blastwave(12, 3, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=1)
    return problems.hd_blast_octant(12, 3, solver=abi.FLUX_RSroe, strict_fp=1, nzones=3.0)


def _setup(case, sim, cfg):
    """cooling tables and this grid's (or slab's) wind cells on an oracle handle; returns the first-step dt limit"""
    if case != "wind3d":
        return None
    from pion_amd import cooling, problems
    sim.set_cooling_tables(*cooling.build_tables(cfg.min_temp, cfg.max_temp))
    _, (idx, st), dt_lim = problems.fill_wind3d(cfg, 12)
    if idx.size:
        sim.set_wind_cells(idx, st)
    return dt_lim


def _worker(rank, world, name, case, nsteps, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["PION_NO_TORCH"] = "1"
        from cpu_backends import CpuSim
        from pion_amd import abi, host_rccl, slab
        be = _orc_backend()
        cfg_g, P = _case(case)
        cfg = slab.slab_config(cfg_g, rank, world)
        periodic = cfg_g.bc_type[4] == abi.BC_PERIODIC
        with host_rccl.HostSim(cfg, 0, rank=rank, world=world, periodic_z=periodic, shm_name=name,
                               backend=be.pion_backend_oracle()) as s:
            o = CpuSim(cfg, "orc", borrowed_handle=be.pion_backend_oracle_handle(s.gpu_handle()))
            dtl = _setup(case, o, cfg)
            s.init(slab.slab_slice(P, cfg_g, rank, world), first_step_dt_limit=dtl)
            n, t, ldt = s.time_int(nsteps)
            q.put((rank, t, s.download(0)))
    except Exception as e:   # noqa: BLE001
        q.put((rank, None, repr(e)))


@pytest.mark.parametrize("case,world", [("glm_periodic", 2), ("hd_octant", 2), ("wind3d", 2), ("glm_periodic", 4), ("hd_octant", 3)])
def test_cpp_time_loop_two_ranks_over_shared_memory(case, world):
    """world = 2: both neighbours of a periodic rank are the same peer; world = 3 / 4: middle ranks with two different
    neighbours, end ranks with one (hd_octant) or the periodic wrap (glm_periodic)"""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_backends import CpuSim
    from pion_amd import driver
    _orc_backend()
    nsteps = 3
    cfg, P = _case(case)
    with CpuSim(cfg, "orc") as o:
        sc = driver.SimControl(o, cfg)
        sc.first_step_dt_limit = _setup(case, o, cfg)
        sc.init(P)
        sc.time_int(nsteps)
        ref, tref = o.download(0), sc.simtime
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import time
    name = "/pion_t%d_%d_%s" % (os.getpid(), time.time_ns() % 1000000007, case)
    procs = [ctx.Process(target=_worker, args=(r, world, name, case, nsteps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, t, A = q.get(timeout=300)
        assert t is not None, A
        res[r] = (t, A)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    nb, nzl = cfg.nbc, cfg.ng[2] // world
    for r in range(world):
        t, A = res[r]
        assert t == tref
        got = A[:, nb:nb + nzl]
        want = ref[:, nb + r * nzl:nb + (r + 1) * nzl]
        assert np.array_equal(got[:, :, nb:-nb, nb:-nb], want[:, :, nb:-nb, nb:-nb]), (case, r)
