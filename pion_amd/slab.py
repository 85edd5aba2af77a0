"""z-slab decomposition over the GPUs of one node (one process per GPU).

Replaces, for this path, the reference's source/decomposition/MCMD_control.cpp:231-309
(slab decomposition along one axis) and the packed MPI halo exchange of
source/comms/comm_mpi.cpp:287-636 / boundaries/MCMD_boundaries.cpp:122-237:

  * rank r owns z in [r*Nz/N, (r+1)*Nz/N) plus nbc ghost planes either side;
  * after every stage, once the local x/y (and physical z) boundaries are filled, the
    nbc on-grid planes next to each internal z face -- full x-y extent INCLUDING the
    x/y ghosts, which reproduces the reference's X->Y->Z corner fill -- are copied to a
    contiguous device buffer and sent to the neighbour with torch.distributed P2P
    (backend nccl = RCCL over xGMI on the GPUs, gloo on CPU for tests);
  * the time step is a 2-double all-reduce(min) (sim_control_MPI.cpp:503-504).
The reference sends Ph only and copies P=Ph locally on the full step; here the array
that was just written (Ph after a half step, P after a full step) is the one exchanged.
"""
from . import abi


def slab_config(cfg_global, rank, world):
    """Per-rank configuration of a z-slab of the global problem."""
    import copy
    if cfg_global.ndim != 3:
        raise ValueError("slab decomposition needs a 3-D grid")
    nz = cfg_global.ng[2]
    if nz % world != 0:
        raise ValueError("NGridZ=%d not divisible by %d ranks" % (nz, world))
    cfg = copy.deepcopy(cfg_global)
    nzl = nz // world
    cfg.ng[2] = nzl
    cfg.xmin[2] = cfg_global.xmin[2] + rank * nzl * cfg_global.dx
    periodic = cfg_global.bc_type[4] == abi.BC_PERIODIC
    if world > 1:
        if periodic or rank > 0:
            cfg.bc_type[4] = abi.BC_SLAB
        if periodic or rank < world - 1:
            cfg.bc_type[5] = abi.BC_SLAB
    return cfg


def slab_slice(P_global, cfg_global, rank, world):
    """The rank's part [nvar][nzl+2nbc][ny_all][nx_all] of a global SoA array (ghost planes
    taken from the global array; they are overwritten by the first boundary update)."""
    nb = cfg_global.nbc
    nzl = cfg_global.ng[2] // world
    z0 = rank * nzl
    return P_global[:, z0:z0 + nzl + 2 * nb].copy()


class _nullctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class SlabComm:
    """Halo exchange + dt reduction for one rank.  `make_buffer(n)` returns a flat float64
    tensor on the device the backend computes on; backend.pack_halo/unpack_halo take
    (which, face, tensor.data_ptr())."""

    def __init__(self, rank, world, periodic, halo_count, device, host_staged=None, loopback=False):
        """loopback: a single rank with periodic z whose z faces are BC_SLAB exchanges with itself
        through the backend (the only way to drive RCCL send/recv on a one-GPU box); the result
        equals the periodic single-domain run.
        host_staged: exchange through pinned host buffers (GPU state, CPU-only backend such as
        gloo: rehearsals of the multi-rank GPU path on a box without RCCL peers).  Default: staged iff
        the device is a GPU and the process group's backend is not nccl."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = rank, world
        self.up = (rank + 1) % world if (periodic or rank < world - 1) else None
        self.down = (rank - 1) % world if (periodic or rank > 0) else None
        self.loopback = bool(loopback and world == 1 and periodic)
        if world == 1:
            self.up = self.down = (rank if self.loopback else None)
        device = torch.device(device)
        mk = lambda: torch.empty(halo_count, dtype=torch.float64, device=device)
        self.send_up, self.send_down, self.recv_up, self.recv_down = mk(), mk(), mk(), mk()
        if host_staged is None:
            host_staged = (device.type == "cuda" and (world > 1 or self.loopback) and dist.is_initialized()
                           and dist.get_backend() != "nccl")
        self.host_staged = bool(host_staged)
        if self.host_staged:
            mkh = lambda: torch.empty(halo_count, dtype=torch.float64).pin_memory()
            self.h_send_up, self.h_send_down, self.h_recv_up, self.h_recv_down = mkh(), mkh(), mkh(), mkh()
        self.dtbuf = torch.empty(2, dtype=torch.float64, device="cpu" if self.host_staged else device)
        self.pending = None
        self.kstream = self.cstream = None

    def use_streams(self, sim):
        """GPU ranks: kernels on one torch stream, halo pack / RCCL / unpack on a second (high
        priority) one, ordered against each other inside the library (pion_gpu_set_comm_stream), so
        that the exchange runs under the interior part of the next stage and the host never blocks."""
        torch = self.torch
        if not self.send_up.is_cuda:
            return
        sim.synchronize()
        self.kstream = torch.cuda.Stream()
        self.cstream = torch.cuda.Stream(priority=-1)
        sim.set_stream(self.kstream.cuda_stream)
        sim.set_comm_stream(self.cstream.cuda_stream)

    def start(self, sim, which):
        """Pack the on-grid planes next to the internal z faces of array `which` and post the
        transfers.  Returns at once; finish() must follow before the ghost planes are read."""
        dist = self.dist
        assert self.pending is None, "previous halo exchange not finished"
        if self.up is None and self.down is None:
            return
        ctx = self.torch.cuda.stream(self.cstream) if self.cstream is not None else _nullctx()
        with ctx:
            if self.up is not None:
                sim.pack_halo(which, 5, self.send_up.data_ptr())      # my top on-grid planes
            if self.down is not None:
                sim.pack_halo(which, 4, self.send_down.data_ptr())    # my bottom on-grid planes
            su, sd, ru, rd = self.send_up, self.send_down, self.recv_up, self.recv_down
            if self.host_staged:
                # pack ran on the comm stream (or, without streams, on the handle's stream)
                if self.cstream is None:
                    sim.synchronize()
                if self.up is not None:
                    self.h_send_up.copy_(self.send_up, non_blocking=True)
                if self.down is not None:
                    self.h_send_down.copy_(self.send_down, non_blocking=True)
                self.torch.cuda.current_stream().synchronize()
                su, sd, ru, rd = self.h_send_up, self.h_send_down, self.h_recv_up, self.h_recv_down
            elif self.cstream is None:
                sim.synchronize()
            # order matters when up == down (2 ranks, periodic): first message = "top" planes
            ops = []
            if self.up is not None:
                ops.append(dist.P2POp(dist.isend, su, self.up))
            if self.down is not None:
                ops.append(dist.P2POp(dist.irecv, rd, self.down))
            if self.down is not None:
                ops.append(dist.P2POp(dist.isend, sd, self.down))
            if self.up is not None:
                ops.append(dist.P2POp(dist.irecv, ru, self.up))
            self.pending = (which, dist.batch_isend_irecv(ops))

    def finish(self, sim):
        if self.pending is None:
            return
        which, works = self.pending
        self.pending = None
        ctx = self.torch.cuda.stream(self.cstream) if self.cstream is not None else _nullctx()
        with ctx:
            for w in works:
                w.wait()        # GPU + RCCL: the comm stream waits, the host does not
            if self.host_staged:
                if self.down is not None:
                    self.recv_down.copy_(self.h_recv_down, non_blocking=True)
                if self.up is not None:
                    self.recv_up.copy_(self.h_recv_up, non_blocking=True)
                if self.cstream is None:
                    self.torch.cuda.current_stream().synchronize()
            elif self.cstream is None and self.recv_down.is_cuda:
                self.torch.cuda.current_stream().synchronize()
            if self.down is not None:
                sim.unpack_halo(which, 4, self.recv_down.data_ptr())  # neighbour's top -> my ZN ghosts
            if self.up is not None:
                sim.unpack_halo(which, 5, self.recv_up.data_ptr())    # neighbour's bottom -> my ZP ghosts

    def exchange(self, sim, which):
        self.start(sim, which)
        self.finish(sim)

    def allreduce_min(self, t_dyn, t_mp):
        if self.world == 1 and not self.loopback:
            return t_dyn, t_mp
        self.dtbuf[0] = t_dyn
        self.dtbuf[1] = t_mp
        self.dist.all_reduce(self.dtbuf, op=self.dist.ReduceOp.MIN)
        v = self.dtbuf.tolist()
        return v[0], v[1]
