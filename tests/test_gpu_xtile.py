"""The full 62-cell x-tile path of the 3-D stage kernel against the ORACLE.

Every other oracle-compared 3-D grid has nx <= 48, i.e. only the packed remainder wavefront of
k_stage_rows2 runs.  Here nx = 70 (one full tile + an 8-cell remainder) and nx = 130 (two full tiles + a
6-cell remainder), ny not a multiple of the rows per wavefront, and data with gradients across every
tile seam.  Strict build: bit-exact.  Fast (benchmarked) build: cell-wise within 1e-11 of each
variable's scale per step, on problems where the reference itself is well conditioned
(tests/test_reference_conditioning.py)."""
import numpy as np
import pytest

from pion_amd import abi, driver, problems
from test_gpu_parity import run_pair, _gpu, _cpu

pytestmark = pytest.mark.gpu

GRIDS = [[130, 9, 20], [70, 14, 12]]


def _case(name, ng, strict):
    if name == "glm_hlld":
        return problems.mhd_blast_generic(ng, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=strict)
    if name == "mhd_hlld":
        return problems.mhd_blast_generic(ng, abi.EQMHD, abi.FLUX_RS_HLLD, strict_fp=strict)
    if name == "glm_hlld_tr":
        return problems.mhd_blast_generic(ng, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=strict, ntracer=1)
    if name == "hd_roe":
        return problems.hd_blast_box(ng, solver=abi.FLUX_RSroe, strict_fp=strict)
    if name == "hd_fvs_tr":
        return problems.hd_blast_box(ng, solver=abi.FLUX_FVS, ntracer=1, strict_fp=strict)
    raise KeyError(name)


CASES = ["glm_hlld", "mhd_hlld", "glm_hlld_tr", "hd_roe", "hd_fvs_tr"]


@pytest.mark.parametrize("ng", GRIDS, ids=lambda g: "x".join(map(str, g)))
@pytest.mark.parametrize("case", CASES)
def test_full_xtile_strict_bitexact_vs_oracle(case, ng):
    cfg, P = _case(case, ng, 1)
    run_pair(cfg, P, 2)


@pytest.mark.parametrize("ng", GRIDS, ids=lambda g: "x".join(map(str, g)))
@pytest.mark.parametrize("case", CASES)
def test_full_xtile_fast_vs_oracle(case, ng):
    """the benchmarked build, cell by cell against the oracle: <= 1e-11 of each variable's scale per step"""
    cfg, P = _case(case, ng, 0)
    run_pair(cfg, P, 3, strict=False, tol=3e-11)


def _norms(a, b, refvec):
    """per-variable L1 / L2 / max norms of a-b over the on-grid cells, relative to refvec
    (the norms of analysis/silocompare/silocompare.cpp:371-430)"""
    nv = a.shape[0]
    d = np.abs(a - b).reshape(nv, -1)
    rv = np.asarray(refvec[:nv]).reshape(-1, 1)
    return (d.mean(axis=1) / rv[:, 0], np.sqrt((d * d).mean(axis=1)) / rv[:, 0], d.max(axis=1) / rv[:, 0])


@pytest.mark.parametrize("eq", [abi.EQGLM, abi.EQMHD])
def test_fast_build_vs_oracle_blast_64cubed_10_steps(eq):
    """SURVEY 8(d) gate for the benchmarked build: per-variable L1 and L2 <= 1e-10 x refvec against the
    oracle after ten steps of the (well-conditioned) MHD blast on 64^3 -- two full x tiles are not needed
    here (nx = 64 = one full tile + 2), the shock crosses several cells."""
    cfg, P = problems.mhd_blast_generic([64, 64, 64], eq, abi.FLUX_RS_HLLD, strict_fp=0)
    nb = cfg.nbc
    with _gpu(cfg) as g, _cpu(cfg) as o:
        sg, so = driver.SimControl(g, cfg), driver.SimControl(o, cfg)
        sg.init(P)
        so.init(P)
        for _ in range(10):
            dg, do = sg.calculate_timestep(), so.calculate_timestep()
            assert abs(dg - do) <= 1e-11 * do
            so.dt = sg.dt
            sg.advance_time()
            so.advance_time()
        a = g.download(0)[:, nb:-nb, nb:-nb, nb:-nb]
        b = o.download(0)[:, nb:-nb, nb:-nb, nb:-nb]
    refvec = [cfg.refvec[v] for v in range(cfg.nvar)]
    l1, l2, mx = _norms(a, b, refvec)
    assert l1.max() <= 1e-10 and l2.max() <= 1e-10, (l1, l2, mx)
    assert mx.max() <= 1e-8, mx


def test_strict_build_bitexact_blast_64cubed():
    """the same problem, parity build: bit for bit after three steps (oracle cost ~2 s/step)"""
    cfg, P = problems.mhd_blast_generic([64, 64, 64], abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=1)
    run_pair(cfg, P, 3)


@pytest.mark.parametrize("strict", [1, 0])
@pytest.mark.parametrize("case", ["hd_roe", "hd_fvs_tr", "glm_hlld"])
def test_rows_per_wavefront_do_not_change_the_result(case, strict, monkeypatch):
    """k_stage_rows2 picks its rows per wavefront per instance (Euler: for three workgroups' LDS per CU; MHD: for
    two; PION_ROWS / PION_ROWS1 override).  The choice only moves work between wavefronts: every interface flux
    is the same function of the same states, so the result is the same bit for bit -- in the fast build too."""
    cfg, P = _case(case, [70, 14, 12], strict)
    out = {}
    for rows in (None, "1", "3"):
        if rows is None:
            monkeypatch.delenv("PION_ROWS", raising=False)
        else:
            monkeypatch.setenv("PION_ROWS", rows)
        with _gpu(cfg) as g:
            sc = driver.SimControl(g, cfg)
            sc.init(P)
            sc.time_int(3)
            out[rows] = (g.download(0), sc.simtime)
    for rows in ("1", "3"):
        assert out[rows][1] == out[None][1]
        assert np.array_equal(out[rows][0], out[None][0]), rows


# ---- 2-D Cartesian grids run the same kernel without its z part (rows marched along y) -----------
GRIDS_2D = [[130, 37], [70, 19], [24, 64]]


@pytest.mark.parametrize("ng", GRIDS_2D, ids=lambda g: "x".join(map(str, g)))
@pytest.mark.parametrize("case", CASES)
def test_2d_rows_kernel_strict_bitexact_vs_oracle(case, ng):
    cfg, P = _case(case, ng, 1)
    run_pair(cfg, P, 3)


@pytest.mark.parametrize("ng", GRIDS_2D[:2], ids=lambda g: "x".join(map(str, g)))
@pytest.mark.parametrize("case", CASES)
def test_2d_rows_kernel_fast_vs_oracle(case, ng):
    cfg, P = _case(case, ng, 0)
    run_pair(cfg, P, 3, strict=False, tol=3e-11)


@pytest.mark.parametrize("case", ["glm_hlld", "hd_roe"])
@pytest.mark.parametrize("rows", ["1", "3", "20"])
def test_2d_rows_per_wavefront_and_cell_kernel_agree(case, rows, monkeypatch):
    """rows per wavefront (PION_ROWS) only move work between wavefronts; PION_ROWS_2D=0 is the cell-per-thread
    kernel: the strict build gives the same bits every way, first-order-in-space grids (OA1) included"""
    out = []
    for env in ({}, {"PION_ROWS": rows}, {"PION_ROWS_2D": "0"}):
        for k in ("PION_ROWS", "PION_ROWS_2D"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        cfg, P = _case(case, [70, 41], 1)
        with _gpu(cfg) as g:
            sc = driver.SimControl(g, cfg)
            sc.init(P)
            sc.time_int(3)
            out.append((g.download(0), sc.simtime))
    for A, t in out[1:]:
        assert t == out[0][1]
        assert np.array_equal(A, out[0][0])


# ---- production-sized launches: many x tiles + the remainder tile, uneven plane chunks, every XCD's share of tiles ----
@pytest.mark.parametrize("case,ng", [("glm_hlld", [200, 140, 150]), ("hd_roe", [330, 70, 100]), ("glm_hlld", [2048, 630]),
                                     ("hd_fvs_tr", [1000, 333])], ids=["glm3d", "hd3d", "glm2d", "hdtr2d"])
def test_rows_kernel_equals_cell_kernel_on_large_grids(case, ng, monkeypatch):
    """the rows kernel (3-D: uneven chunks, per-XCD tile map; 2-D: rows marched along y) against the cell-per-thread
    kernel on grids with hundreds of thousands of wavefront tiles, strict build: bit for bit after three steps, and
    the time steps of the fused reduction equal to k_dt's"""
    out = []
    for env in ({}, {"PION_STAGE_KERNEL": "cell", "PION_ROWS_2D": "0"}):
        for k in ("PION_STAGE_KERNEL", "PION_ROWS_2D"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        cfg, P = _case(case, ng, 1)
        with _gpu(cfg) as g:
            sc = driver.SimControl(g, cfg)
            sc.init(P)
            dts = []
            for _ in range(3):
                dts.append(sc.calculate_timestep())
                sc.advance_time()
            out.append((g.download(0), dts))
        del P
    assert out[0][1] == out[1][1]
    assert np.array_equal(out[0][0], out[1][0])


# ---- cylindrical (z,R) 2-D grids: the CYL instance of the rows kernel (geometry in slopes, edge states, divergence and
# source terms along R) against the oracle and against the cell-per-thread kernel -------------------------------------
def _cyl_case(kind, n, strict):
    if kind == "hd_roe":
        return problems.blast_axi2d(n, abi.EQEUL, abi.FLUX_RSroe, strict_fp=strict)
    if kind == "hd_roe_tr_hcorr":
        return problems.blast_axi2d(n, abi.EQEUL, abi.FLUX_RSroe, ntracer=1, artvisc=abi.AV_HCORR_FKJ98, strict_fp=strict)
    if kind == "hd_fvs":
        return problems.blast_axi2d(n, abi.EQEUL, abi.FLUX_FVS, strict_fp=strict)
    if kind == "mhd_hlld":
        return problems.blast_axi2d(n, abi.EQMHD, abi.FLUX_RS_HLLD, strict_fp=strict)
    if kind == "glm_hlld_tr":
        return problems.blast_axi2d(n, abi.EQGLM, abi.FLUX_RS_HLLD, ntracer=1, strict_fp=strict)
    if kind == "glm_roe":
        return problems.blast_axi2d(n, abi.EQGLM, abi.FLUX_RSroe, strict_fp=strict)
    if kind == "glm_jet":
        cfg, P, _ = problems.jet_axi2d(n, strict_fp=strict)
        return cfg, P
    raise KeyError(kind)


CYL_CASES = ["hd_roe", "hd_roe_tr_hcorr", "hd_fvs", "mhd_hlld", "glm_hlld_tr", "glm_roe"]


@pytest.mark.parametrize("n", [24, 140], ids=["24x12", "140x70"])
@pytest.mark.parametrize("kind", CYL_CASES)
def test_cyl_rows_kernel_strict_bitexact_vs_oracle(kind, n):
    """one x tile and three (two full + remainder), 12 / 70 rows (not multiples of the 16 rows per wavefront)"""
    cfg, P = _cyl_case(kind, n, 1)
    run_pair(cfg, P, 3)


@pytest.mark.parametrize("kind", CYL_CASES)
def test_cyl_rows_kernel_fast_vs_oracle(kind):
    cfg, P = _cyl_case(kind, 140, 0)
    run_pair(cfg, P, 3, strict=False, tol=3e-11)


@pytest.mark.parametrize("kind", ["hd_roe", "glm_hlld_tr"])
@pytest.mark.parametrize("rows", ["1", "3", "20"])
def test_cyl_rows_per_wavefront_and_cell_kernel_agree(kind, rows, monkeypatch):
    out = []
    for env in ({}, {"PION_ROWS": rows}, {"PION_ROWS_2D": "0"}):
        for k in ("PION_ROWS", "PION_ROWS_2D"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        cfg, P = _cyl_case(kind, 76, 1)
        with _gpu(cfg) as g:
            sc = driver.SimControl(g, cfg)
            sc.init(P)
            sc.time_int(3)
            out.append((g.download(0), sc.simtime))
    for A, t in out[1:]:
        assert t == out[0][1]
        assert np.array_equal(A, out[0][0])


@pytest.mark.parametrize("kind", ["glm_hlld_tr", "hd_roe"])
def test_cyl_rows_kernel_equals_cell_kernel_on_a_large_grid(kind, monkeypatch):
    """2048 x 1024 axisymmetric grid: thousands of tiles, every XCD's share, the fused time-step reduction"""
    out = []
    for env in ({}, {"PION_STAGE_KERNEL": "cell", "PION_ROWS_2D": "0"}):
        for k in ("PION_STAGE_KERNEL", "PION_ROWS_2D"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        cfg, P = _cyl_case(kind, 2048, 1)
        with _gpu(cfg) as g:
            sc = driver.SimControl(g, cfg)
            sc.init(P)
            dts = []
            for _ in range(3):
                dts.append(sc.calculate_timestep())
                sc.advance_time()
            out.append((g.download(0), dts))
        del P
    assert out[0][1] == out[1][1]
    assert np.array_equal(out[0][0], out[1][0])


def test_cyl_jet_with_internal_boundary_strict_vs_oracle():
    cfg, P, (radius, state) = problems.jet_axi2d(96, strict_fp=1)
    res = []
    for mk in (lambda: _cpu(cfg), lambda: _gpu(cfg)):
        with mk() as s:
            s.set_jet(radius, state)
            sc = driver.SimControl(s, cfg)
            sc.init(P)
            sc.time_int(4)
            res.append((s.download(0), sc.simtime))
    assert res[0][1] == res[1][1]
    assert np.array_equal(res[0][0], res[1][0])
