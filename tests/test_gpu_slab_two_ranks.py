"""Two ranks, GPU state, one device: the whole N > 1 product path (slab configuration, split stages,
compute + comm streams, halo pack / unpack, dt reduction) with the transport replaced by gloo over
pinned host buffers -- no RCCL peer exists on a one-GPU box.  Must reproduce the single-domain GPU
run bit for bit (strict build)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, case, nsteps, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from pion_amd import abi, driver, lib, slab
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg_g, P = _case(case)
    cfg = slab.slab_config(cfg_g, rank, world)
    periodic = cfg_g.bc_type[4] == abi.BC_PERIODIC
    with lib.GpuSim(cfg, 0) as g:
        comm = slab.SlabComm(rank, world, periodic, g.halo_count(), torch.device("cuda", 0))
        assert comm.host_staged
        comm.use_streams(g)
        sc = driver.SimControl(g, cfg, comm=comm)
        sc.first_step_dt_limit = _setup(case, g, cfg)
        sc.init(slab.slab_slice(P, cfg_g, rank, world))
        sc.time_int(nsteps)
        q.put((rank, sc.simtime, g.download(0)))
    dist.barrier()
    dist.destroy_process_group()


def _setup(case, sim, cfg):
    """cooling tables, this grid's (or slab's) wind cells; returns the first-step dt limit"""
    if case != "wind3d":
        return None
    from pion_amd import cooling, problems
    sim.set_cooling_tables(*cooling.build_tables(cfg.min_temp, cfg.max_temp))
    _, (idx, st), dt_lim = problems.fill_wind3d(cfg, 16)
    if idx.size:
        sim.set_wind_cells(idx, st)
    return dt_lim


def _case(case):
    from pion_amd import abi, problems
    if case == "wind3d":
        cfg, P, _, _ = problems.wind3d(16, strict_fp=1)
        return cfg, P
    if case == "glm_periodic":
        return problems.mhd_blastwave(16, 3, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=1)
    return problems.hd_blast_octant(16, 3, solver=abi.FLUX_RSroe, strict_fp=1, nzones=3.0)


@pytest.mark.parametrize("case", ["glm_periodic", "hd_octant", "wind3d"])
def test_two_gpu_ranks_match_single_domain(case):
    import torch.multiprocessing as mp
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
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, nsteps, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, t, A = q.get(timeout=300)
        got[r] = (t, A)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    nb, nzl = cfg.nbc, cfg.ng[2] // 2
    for r in range(2):
        t, A = got[r]
        assert t == tref
        # the rank's on-grid planes (x/y ghosts included) against the same planes of the single domain
        want = ref[:, nb + r * nzl: nb + (r + 1) * nzl]
        got_r = A[:, nb:nb + nzl]
        assert np.array_equal(got_r[:, :, nb:-nb, nb:-nb], want[:, :, nb:-nb, nb:-nb]), (
            case, r, (got_r != want).sum())
