"""world_size-2 test of the slab decomposition on CPU (gloo): two processes, each driving the
oracle on a z-slab with halo exchange + dt all-reduce, must reproduce the single-domain run
bit for bit (the reference's own expectation for its MPI build, solver_eqn_base.cpp:46-48)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, case, nsteps, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from cpu_backends import CpuSim
    from pion_amd import abi, driver, problems, slab
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg_g, P = _case(case)
    cfg = slab.slab_config(cfg_g, rank, world)
    periodic = cfg_g.bc_type[4] == abi.BC_PERIODIC
    with CpuSim(cfg, "orc") as o:
        comm = slab.SlabComm(rank, world, periodic, o.halo_count(), torch.device("cpu"))
        sc = driver.SimControl(o, cfg, comm=comm)
        sc.init(slab.slab_slice(P, cfg_g, rank, world))
        sc.time_int(nsteps)
        q.put((rank, sc.simtime, o.download(0)))
    dist.barrier()
    dist.destroy_process_group()


def _case(case):
    from pion_amd import abi, problems
    if case == "glm_periodic":
        return problems.mhd_blastwave(12, 3, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=1)
    return problems.hd_blast_octant(12, 3, solver=abi.FLUX_RSroe, strict_fp=1, nzones=3.0)


@pytest.mark.parametrize("case", ["glm_periodic", "hd_octant"])
def test_two_rank_slab_matches_single_domain(case):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_backends import CpuSim
    from pion_amd import driver
    nsteps = 3
    cfg, P = _case(case)
    with CpuSim(cfg, "orc") as o:
        sc = driver.SimControl(o, cfg)
        sc.init(P)
        sc.time_int(nsteps)
        ref, tref = o.download(0), sc.simtime
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, nsteps, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, t, A = q.get(timeout=300)
        res[r] = (t, A)
    for p in procs:
        p.join(timeout=60)
    nb, nzl = cfg.nbc, cfg.ng[2] // 2
    for r in range(2):
        t, A = res[r]
        assert t == tref
        got = A[:, nb:nb + nzl]
        want = ref[:, nb + r * nzl:nb + (r + 1) * nzl]
        assert np.array_equal(got[:, :, nb:-nb, nb:-nb], want[:, :, nb:-nb, nb:-nb]), (case, r)
