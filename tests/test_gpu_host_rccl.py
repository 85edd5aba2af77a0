"""The C++ RCCL path of libpion_host (pion_host::slab_comm_rccl driven by pion_host::sim_control_gpu):
ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the communication stream and
ncclAllReduce(ncclMin) on the device-resident time-step minima, no Python in the time loop.

One-GPU box: a one-rank communicator whose periodic z faces are slab faces (the rank is its own
neighbour) -- the N > 1 code path with the real transport calls.  The result must equal the periodic
single-domain run bit for bit (strict build).  Child process under a time limit, so that a transport hang
cannot take the test session with it."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(case, nsteps, q):
    sys.path.insert(0, ROOT)
    from pion_amd import abi, host_rccl, problems
    if case == "glm":
        cfg, P = problems.mhd_blastwave(16, 3, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=1)
    else:
        cfg, P = problems.mhd_blast_generic([70, 12, 16], abi.EQMHD, abi.FLUX_RS_HLLD, strict_fp=1)
    cfg.bc_type[4] = cfg.bc_type[5] = abi.BC_SLAB
    uid = host_rccl.new_unique_id()
    with host_rccl.HostSim(cfg, 0, rank=0, world=1, periodic_z=True, unique_id=uid) as s:
        s.init(P)
        n, t, ldt = s.time_int(nsteps)
        q.put((n, t, ldt, s.download(0)))


@pytest.mark.parametrize("case", ["glm", "mhd_xtile"])
def test_cpp_rccl_self_exchange_equals_periodic(case):
    import torch.multiprocessing as mp
    from pion_amd import abi, driver, lib, problems
    nsteps = 3
    if case == "glm":
        cfg, P = problems.mhd_blastwave(16, 3, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=1)
    else:
        cfg, P = problems.mhd_blast_generic([70, 12, 16], abi.EQMHD, abi.FLUX_RS_HLLD, strict_fp=1)
    with lib.GpuSim(cfg, 0) as g:
        sc = driver.SimControl(g, cfg)
        sc.init(P)
        sc.time_int(nsteps)
        ref, tref, dtref = g.download(0), sc.simtime, sc.last_dt
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(case, nsteps, q))
    p.start()
    try:
        n, t, ldt, A = q.get(timeout=240)
    except Exception:
        p.kill()
        p.join()
        pytest.fail("C++ RCCL loopback worker produced nothing within 240 s (exit code %s)" % p.exitcode)
    p.join(timeout=60)
    assert p.exitcode == 0
    nb = cfg.nbc
    assert n == nsteps and t == tref and ldt == dtref
    assert np.array_equal(A[:, nb:-nb, nb:-nb, nb:-nb], ref[:, nb:-nb, nb:-nb, nb:-nb])
