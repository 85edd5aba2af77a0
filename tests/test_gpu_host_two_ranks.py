"""The C++ host time loop with TWO RANKS on the GPU box: pion_host::sim_control_gpu + split stages + two streams +
request_min / allreduce_min, as two processes.

  * one GPU (always runs): transport pion_host::slab_comm_shm (host-staged through POSIX shared memory) -- both ranks
    share device 0; GLM-MHD periodic, HD octant, Wind3D; strict build, bit for bit against the single-domain run;
  * two or more GPUs (skipped on a one-GPU box): the SAME cases over pion_host::slab_comm_rccl, one rank per
    device -- the transport `bench.py --gpus N` uses by default (it had never run with N > 1).
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(case):
    from pion_amd import abi, problems
    if case == "wind3d":
        cfg, P, _, _ = problems.wind3d(16, strict_fp=1)
        return cfg, P
    if case == "glm_periodic":
        return problems.mhd_blastwave(16, 3, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=1)
    return problems.hd_blast_octant(16, 3, solver=abi.FLUX_RSroe, strict_fp=1, nzones=3.0)


def _setup(case, sim, cfg):
    if case != "wind3d":
        return None
    from pion_amd import cooling, problems
    sim.set_cooling_tables(*cooling.build_tables(cfg.min_temp, cfg.max_temp))
    _, (idx, st), dt_lim = problems.fill_wind3d(cfg, 16)
    if idx.size:
        sim.set_wind_cells(idx, st)
    return dt_lim


def _worker(rank, world, transport, token, case, nsteps, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from pion_amd import abi, host_rccl, lib, slab
        cfg_g, P = _case(case)
        cfg = slab.slab_config(cfg_g, rank, world)
        periodic = cfg_g.bc_type[4] == abi.BC_PERIODIC
        if transport == "shm":
            kw = dict(shm_name=token)
            device = 0
        else:
            kw = dict(unique_id=token)
            device = rank
        with host_rccl.HostSim(cfg, device, rank=rank, world=world, periodic_z=periodic, **kw) as s:
            g = lib.GpuSim(cfg, device, borrowed_handle=s.gpu_handle())
            dtl = _setup(case, g, cfg)
            s.init(slab.slab_slice(P, cfg_g, rank, world), first_step_dt_limit=dtl)
            n, t, ldt = s.time_int(nsteps)
            q.put((rank, t, s.download(0)))
    except Exception as e:   # noqa: BLE001
        q.put((rank, None, repr(e)))


def _run(case, transport, token):
    import multiprocessing as mp
    from pion_amd import driver, lib
    nsteps = 3
    cfg, P = _case(case)
    with lib.GpuSim(cfg, 0) as g:
        sc = driver.SimControl(g, cfg)
        sc.first_step_dt_limit = _setup(case, g, cfg)
        sc.init(P)
        sc.time_int(nsteps)
        ref, tref = g.download(0), sc.simtime
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, transport, token, case, nsteps, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(2):
            r, t, A = q.get(timeout=300)
            assert t is not None, A
            got[r] = (t, A)
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    nb, nzl = cfg.nbc, cfg.ng[2] // 2
    for r in range(2):
        t, A = got[r]
        assert t == tref
        want = ref[:, nb + r * nzl: nb + (r + 1) * nzl]
        got_r = A[:, nb:nb + nzl]
        assert np.array_equal(got_r[:, :, nb:-nb, nb:-nb], want[:, :, nb:-nb, nb:-nb]), (case, r, (got_r != want).sum())


@pytest.mark.parametrize("case", ["glm_periodic", "hd_octant", "wind3d"])
def test_cpp_loop_two_ranks_one_gpu_shared_memory_transport(case):
    import time
    _run(case, "shm", "/pion_g%d_%d_%s" % (os.getpid(), time.time_ns() % 1000000007, case))


def _ngpu():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:   # noqa: BLE001
        return 0


@pytest.mark.skipif(_ngpu() < 2, reason="needs two GPUs: the RCCL transport between two devices")
@pytest.mark.parametrize("case", ["glm_periodic", "hd_octant"])
def test_cpp_loop_two_ranks_two_gpus_rccl(case):
    """periodic GLM: with two ranks both neighbours are the same peer (the message-order case of
    slab_comm_rccl::start); HD octant: the end ranks have one neighbour and a physical z face each"""
    from pion_amd import host_rccl
    _run(case, "rccl", host_rccl.new_unique_id())
