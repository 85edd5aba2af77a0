"""k_bc_all -- every external face of any mix of boundary types in one launch, each ghost cell resolving its chain of
source cells down the axes -- against the per-face launch sequence X -> Y -> Z (k_bc_face, PION_FUSE_BC=0), which is
the reference's order (assign_update_bcs.cpp:185-252) and is itself pinned to the reference's boundary updaters by the
whole-step fixtures (tests/test_gpu_golden.py).  Random states everywhere (ghosts included), random mixes of face
types in 1-D, 2-D and 3-D for all three equation sets: the ghost cells must come out bit for bit the same -- corner
and edge ghosts (chains of two and three faces), the psi mirror rule of outflow / one-way faces through corners, the
one-way clamp behind a reflecting flip, inflow / fixed constants under other faces' operations, slab faces left alone."""
import numpy as np
import pytest

from pion_amd import abi

pytestmark = pytest.mark.gpu

TYPES = ["periodic", "outflow", "one-way-outflow", "reflecting", "inflow", "fixed"]


def _run(cfg, P, fuse, monkeypatch):
    from pion_amd import lib
    if fuse:
        monkeypatch.delenv("PION_FUSE_BC", raising=False)
    else:
        monkeypatch.setenv("PION_FUSE_BC", "0")
    with lib.GpuSim(cfg, 0) as g:
        g.upload(P)
        g.update_bcs(0.0, 2, 2, assign=1)      # assignment: per-face sequence in both modes (captures inflow / fixed states)
        A0 = g.download(0)
        # scramble the ghosts again (keep the on-grid cells), then update WITHOUT assignment: this is the launch under test
        Q = P.copy()
        nb = cfg.nbc
        sl = tuple([slice(None)] + [slice(nb, -nb) if a < cfg.ndim else slice(None) for a in (2, 1, 0)])
        Q[sl] = A0[sl]
        g.upload(Q)
        g.update_bcs(0.0, 2, 2, assign=0)
        return g.download(0)


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("eq", [abi.EQEUL, abi.EQMHD, abi.EQGLM])
def test_one_launch_equals_face_sequence(eq, seed, monkeypatch):
    rng = np.random.default_rng(1000 * eq + seed)
    ndim = [3, 3, 2, 3, 2, 1][seed % 6]
    ng = [int(rng.integers(5, 12)) for _ in range(ndim)]
    bcs = []
    for a in range(ndim):
        if rng.uniform() < 0.25:
            bcs += ["periodic", "periodic"]     # (periodic faces come in pairs)
        else:
            bcs += [TYPES[int(rng.integers(1, len(TYPES)))], TYPES[int(rng.integers(1, len(TYPES)))]]
    if all(b == "periodic" for b in bcs):
        bcs[0] = bcs[1] = "outflow"             # (all-periodic grids take k_bc_periodic_all)
    solver = abi.FLUX_RSroe if eq == abi.EQEUL else abi.FLUX_RS_HLLD
    ntr = int(rng.integers(0, 2))
    cfg = abi.make_config(ndim, ng, eq, solver, ntracer=ntr, xmax=(1.0, 1.0, 1.0), bcs=bcs, strict_fp=1)
    shape = (cfg.nvar,) + tuple((ng[a] + 2 * cfg.nbc) if a < ndim else 1 for a in (2, 1, 0))
    P = rng.normal(0.0, 1.0, shape)
    P[0] = np.abs(P[0]) + 0.1
    P[1] = np.abs(P[1]) + 0.1
    a = _run(cfg, P, True, monkeypatch)
    b = _run(cfg, P, False, monkeypatch)
    assert np.array_equal(a, b), (bcs, ng, int((a != b).sum()))
