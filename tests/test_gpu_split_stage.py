"""pion_gpu_stage_part: interior + z-boundary parts with the z halo arriving in between (on a
second stream) must give bit for bit the whole-stage result.  One process, one GPU: the slab's two
z neighbours are the slab itself (periodic), so the 'transfer' is a device copy on the comm stream."""
import copy

import numpy as np
import pytest

from pion_amd import abi, driver, problems

pytestmark = pytest.mark.gpu


class SelfComm:
    """SlabComm's call shape (start/finish/allreduce_min) for a periodic slab whose neighbour is itself."""

    def __init__(self, sim, two_streams):
        import torch
        self.torch = torch
        n = sim.halo_count()
        mk = lambda: torch.empty(n, dtype=torch.float64, device="cuda:0")
        self.top, self.bottom = mk(), mk()
        self.pending = None
        self.kstream = self.cstream = None
        if two_streams:
            sim.synchronize()
            self.kstream, self.cstream = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
            sim.set_stream(self.kstream.cuda_stream)
            sim.set_comm_stream(self.cstream.cuda_stream)

    def start(self, sim, which):
        assert self.pending is None
        sim.pack_halo(which, 5, self.top.data_ptr())
        sim.pack_halo(which, 4, self.bottom.data_ptr())
        self.pending = which

    def finish(self, sim):
        if self.pending is None:
            return
        which, self.pending = self.pending, None
        sim.unpack_halo(which, 4, self.top.data_ptr())      # "neighbour's" top planes -> my ZN ghosts
        sim.unpack_halo(which, 5, self.bottom.data_ptr())

    def allreduce_min(self, a, b):
        return a, b


def _run(cfg, P, nsteps, comm_mode):
    from pion_amd import lib
    with lib.GpuSim(cfg, 0) as g:
        comm = None if comm_mode is None else SelfComm(g, comm_mode == "streams")
        sc = driver.SimControl(g, cfg, comm=comm)
        sc.init(P)
        dts = []
        for _ in range(nsteps):
            dts.append(sc.calculate_timestep())
            sc.advance_time()
        sc.finish_halo()
        return dts, g.download(0)


@pytest.mark.parametrize("strict", [1, 0])
@pytest.mark.parametrize("mode", ["one_stream", "streams"])
@pytest.mark.parametrize("case", ["glm_hlld_20", "glm_hlld_nz5", "glm_hlld_nz4", "mhd_hll_hcorr", "hd_roe"])
def test_split_stage_equals_whole_stage(case, mode, strict):
    if case.startswith("glm"):
        cfg, _ = problems.mhd_blastwave(4, 3, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=strict)
        cfg.ng[0], cfg.ng[1] = 70, 9                      # two x tiles, ragged row groups
        cfg.ng[2] = {"glm_hlld_20": 20, "glm_hlld_nz5": 5, "glm_hlld_nz4": 4}[case]
        cfg.dx = 1.0 / 70
        cfg.xmin[1], cfg.xmin[2] = -4.5 / 70, -0.5 * cfg.ng[2] / 70   # keep the hot sphere on the grid
        P = problems.fill_mhd_blastwave(cfg)
    elif case == "mhd_hll_hcorr":
        cfg, P = problems.mhd_blastwave(12, 3, abi.EQMHD, abi.FLUX_RS_HLL, strict_fp=strict)
        cfg.artvisc = abi.AV_HCORRECTION                  # not split: everything in the z-boundary call
    else:
        cfg0, P0 = problems.hd_blast_octant(12, 3, solver=abi.FLUX_RSroe, strict_fp=strict, nzones=3.0)
        for f in range(6):
            cfg0.bc_type[f] = abi.BC_PERIODIC
        cfg, P = cfg0, P0
    dts_w, whole = _run(cfg, P, 3, None)
    cfg_s = copy.deepcopy(cfg)
    cfg_s.bc_type[4] = cfg_s.bc_type[5] = abi.BC_SLAB
    dts_s, split = _run(cfg_s, P, 3, mode)
    assert dts_w == dts_s
    assert np.array_equal(whole, split), "%d values differ" % (whole != split).sum()
