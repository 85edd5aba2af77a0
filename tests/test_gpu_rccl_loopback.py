"""RCCL on a one-GPU box: a single rank (backend nccl, world_size 1) whose periodic z faces are
BC_SLAB and exchange with the rank itself -- batch_isend_irecv send/recv to self on the comm
stream, event-ordered against the split stage kernels exactly as between two GPUs.  The result must
equal the periodic single-domain run bit for bit (strict build).  Runs in a child process under a
time limit so that a transport hang cannot take the test session with it."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(port, nsteps, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from pion_amd import abi, driver, lib, problems, slab
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    cfg, P = problems.mhd_blastwave(16, 3, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=1)
    cfg.bc_type[4] = cfg.bc_type[5] = abi.BC_SLAB
    with lib.GpuSim(cfg, 0) as g:
        comm = slab.SlabComm(0, 1, True, g.halo_count(), torch.device("cuda", 0), loopback=True)
        assert comm.loopback and not comm.host_staged and comm.up == 0 and comm.down == 0
        comm.use_streams(g)
        sc = driver.SimControl(g, cfg, comm=comm)
        sc.init(P)
        sc.time_int(nsteps)
        q.put((sc.simtime, g.download(0)))
    dist.destroy_process_group()


def test_rccl_self_exchange_equals_periodic():
    import torch.multiprocessing as mp
    from pion_amd import abi, driver, lib, problems
    nsteps = 3
    cfg, P = problems.mhd_blastwave(16, 3, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=1)
    with lib.GpuSim(cfg, 0) as g:
        sc = driver.SimControl(g, cfg)
        sc.init(P)
        sc.time_int(nsteps)
        ref, tref = g.download(0), sc.simtime
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(31500 + (os.getpid() % 2000), nsteps, q))
    p.start()
    try:
        t, A = q.get(timeout=240)
    except Exception:
        p.kill()
        p.join()
        pytest.fail("RCCL loopback worker produced nothing within 240 s (exit code %s)" % p.exitcode)
    p.join(timeout=60)
    assert p.exitcode == 0
    nb = cfg.nbc
    assert t == tref
    assert np.array_equal(A[:, nb:-nb, nb:-nb, nb:-nb], ref[:, nb:-nb, nb:-nb, nb:-nb])
