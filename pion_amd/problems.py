"""Synthetic initial conditions shaped like the reference's test problems.

Every function returns (cfg, P) with P a float64 array [nvar][nz_all][ny_all][nx_all]
(ghost cells included, to be filled by update_bcs) -- the SoA layout of
include/pion_gpu.h.  The parameter values come from the reference's shipped
parameter files (test_problems/*/params_*.txt); the IC formulas follow
source/ics/blast_wave.cpp:626-685 and source/ics/basic_tests.cpp:736 without
the sub-cell volume blending (these are synthetic workloads, not fixtures).
"""
import math

import numpy as np

from . import abi


def cell_centres(cfg):
    """Physical cell-centre coordinates (x[nx_all], y[ny_all], z[nz_all]) -- CI.get_dpos,
    source/grid/cell_interface.cpp:506-512."""
    out = []
    for a in range(3):
        nb = cfg.nbc if a < cfg.ndim else 0
        i = np.arange(-nb, cfg.ng[a] + nb)
        out.append(cfg.xmin[a] + (2 * i + 1) * (0.5 * cfg.dx))
    return out


def alloc(cfg):
    n = abi.ng_all(cfg)
    return np.zeros((cfg.nvar, n[2], n[1], n[0]))


def mesh(cfg):
    x, y, z = cell_centres(cfg)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    return X, Y, Z


def mhd_blastwave(n, ndim=3, eqntype=abi.EQGLM, solver=abi.FLUX_RS_HLLD, strict_fp=0):
    """M1: Stone's MHD blast wave (test_problems/MHD_Blastwave2D/
    params_MHD_blastwave2D_UG_B010_n256.txt) extruded to `ndim` dimensions on [-1/2,1/2]^ndim,
    periodic, gamma 5/3, CFL 0.24, FKJ98 eta 0.1; code-unit B = (1/sqrt2, 1/sqrt2, 0)."""
    ng = [n] * ndim
    cfg = abi.make_config(ndim, ng, eqntype, solver, artvisc=abi.AV_FKJ98_1D, etav=0.1,
                          gamma=5.0 / 3.0, cfl=0.24, xmin=(-0.5, -0.5, -0.5), xmax=(0.5, 0.5, 0.5),
                          bcs=["periodic"] * (2 * ndim),
                          refvec=[1.0, 0.1, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0], strict_fp=strict_fp)
    return cfg, fill_mhd_blastwave(cfg)


def fill_mhd_blastwave(cfg):
    """IC of mhd_blastwave for any (global or slab) configuration; built plane-wise so that a
    512^3 slab does not need several full-size temporaries."""
    P = alloc(cfg)
    x, y, z = cell_centres(cfg)
    P[abi.RO] = 1.0
    P[abi.BX] = 1.0 / math.sqrt(2.0)
    P[abi.BY] = 1.0 / math.sqrt(2.0)
    xy2 = x[None, :] ** 2 + (y[:, None] ** 2 if cfg.ndim > 1 else 0.0)
    for k in range(z.size):
        r2 = xy2 + (z[k] ** 2 if cfg.ndim > 2 else 0.0)
        P[abi.PG, k] = np.where(r2 < 0.1 * 0.1, 10.0, 0.1)
    return P


def mhd_smooth(n, ndim=3, eqntype=abi.EQGLM, solver=abi.FLUX_RS_HLLD, bcs=None, strict_fp=1,
               artvisc=abi.AV_FKJ98_1D):
    """M1-smooth: rho = 1+0.2 sin(2 pi x) cos(2 pi y), p=1, v=(0.5,0.3,0.1),
    B=(1,0.5,0.25)/sqrt(4 pi), psi a small smooth field."""
    ng = [n] * ndim
    cfg = abi.make_config(ndim, ng, eqntype, solver, artvisc=artvisc, etav=0.1,
                          gamma=5.0 / 3.0, cfl=0.24, xmin=(0.0, 0.0, 0.0), xmax=(1.0, 1.0, 1.0),
                          bcs=bcs or ["periodic"] * (2 * ndim), strict_fp=strict_fp)
    P = alloc(cfg)
    X, Y, Z = mesh(cfg)
    tp = 2.0 * math.pi
    P[abi.RO] = 1.0 + 0.2 * np.sin(tp * X) * np.cos(tp * Y)
    P[abi.PG] = 1.0 + 0.1 * np.cos(tp * (X + Z))
    P[abi.VX] = 0.5 + 0.1 * np.sin(tp * Y)
    P[abi.VY] = 0.3 - 0.1 * np.sin(tp * Z)
    P[abi.VZ] = 0.1 + 0.1 * np.sin(tp * X)
    s = 1.0 / math.sqrt(4.0 * math.pi)
    P[abi.BX] = s * (1.0 + 0.1 * np.sin(tp * Y))
    P[abi.BY] = s * (0.5 + 0.1 * np.sin(tp * Z))
    P[abi.BZ] = s * (0.25 + 0.1 * np.sin(tp * X))
    if eqntype == abi.EQGLM:
        P[abi.SI] = 0.01 * np.sin(tp * (X + Y))
    return cfg, P


def hd_blast_octant(n, ndim=3, solver=abi.FLUX_RSroe, ntracer=0, artvisc=abi.AV_FKJ98_1D,
                    nzones=4.0, strict_fp=0):
    """M2: octant Sedov blast, test_problems/blastwave_crt3d/params_BWcrt3D_Octant_NR064.txt
    scaled to n cells: reflecting negative faces, outflow positive faces, Euler gamma 5/3,
    Roe-CV + FKJ98 0.1, cgs-like units."""
    ng = [n] * ndim
    L = 3.086e18
    bcs = []
    for a in range(ndim):
        bcs += ["reflecting", "outflow"]
    rho0, p0 = 2.338e-24, 1.38e-13
    cfg = abi.make_config(ndim, ng, abi.EQEUL, solver, ntracer=ntracer, artvisc=artvisc, etav=0.1,
                          gamma=5.0 / 3.0, cfl=0.3, xmin=(0.0, 0.0, 0.0), xmax=(L, L, L), bcs=bcs,
                          refvec=[rho0, p0, 1e6, 1e6, 1e6] + [1.0] * ntracer, strict_fp=strict_fp)
    P = alloc(cfg)
    X, Y, Z = mesh(cfg)
    r2 = X * X + (Y * Y if ndim > 1 else 0.0) + (Z * Z if ndim > 2 else 0.0)
    rb = nzones * cfg.dx
    P[abi.RO] = rho0
    # 1e51 erg in the full sphere
    vol = 4.0 / 3.0 * math.pi * rb ** 3
    pin = 1.0e51 * (cfg.gamma - 1.0) / vol
    P[abi.PG] = np.where(r2 < rb * rb, pin, p0)
    for t in range(ntracer):
        P[5 + t] = np.where(r2 < rb * rb, 1.0, 0.0)
    return cfg, P


def double_mach_reflection(nx, solver=abi.FLUX_RSroe, strict_fp=0):
    """test_problems/double_Mach_reflection/params_DMR_n130.txt scaled to nx cells in x
    (aspect 3.25:1): IC_basic_tests::setup_DoubleMachRef (ics/basic_tests.cpp:736)."""
    ny = max(4, int(round(nx / 3.25)))
    dx = 3.25 / nx
    cfg = abi.make_config(2, [nx, ny], abi.EQEUL, solver, artvisc=abi.AV_FKJ98_1D, etav=0.1,
                          gamma=1.4, cfl=0.4, dx=dx, xmin=(0.0, 0.0, 0.0),
                          bcs=["inflow", "outflow", "reflecting", "DMR"], bc_dmach2=1,
                          refvec=[1.0] * 5, strict_fp=strict_fp)
    P = alloc(cfg)
    X, Y, Z = mesh(cfg)
    x0 = 1.0 / 6.0 + Y / math.tan(math.pi / 3.0)
    post = X <= x0
    P[abi.RO] = np.where(post, 8.0, 1.4)
    P[abi.PG] = np.where(post, 116.5, 1.0)
    P[abi.VX] = np.where(post, 7.14470958, 0.0)
    P[abi.VY] = np.where(post, -4.125, 0.0)
    return cfg, P


def random_states(rng, n, eqntype, ntracer=0, kind="mixed"):
    """n physically admissible primitive state pairs (left, right) for interface-flux tests,
    including degenerate cases (equal states, Bx=0, B_t=0, supersonic either way)."""
    base = {abi.EQEUL: 5, abi.EQMHD: 8, abi.EQGLM: 9}[eqntype]
    nv = base + ntracer
    L = np.zeros((n, nv))
    R = np.zeros((n, nv))
    for S in (L, R):
        S[:, abi.RO] = np.exp(rng.uniform(-2, 2, n))
        S[:, abi.PG] = np.exp(rng.uniform(-3, 3, n))
        S[:, abi.VX:abi.VZ + 1] = rng.normal(0, 1.5, (n, 3))
        if base >= 8:
            S[:, abi.BX:abi.BZ + 1] = rng.normal(0, 1.0, (n, 3))
        if base == 9:
            S[:, abi.SI] = rng.normal(0, 0.3, n)
        for t in range(ntracer):
            S[:, base + t] = rng.uniform(-0.1, 1.2, n)
    k = n // 10
    if k > 0:
        R[0:k] = L[0:k]                                   # identical states
        R[k:2 * k] = L[k:2 * k] * (1 + 1e-13)             # equalD-close states
        L[2 * k:3 * k, abi.VX] = 8.0                      # supersonic to the right
        R[2 * k:3 * k, abi.VX] = 7.5
        L[3 * k:4 * k, abi.VX] = -8.0                     # supersonic to the left
        R[3 * k:4 * k, abi.VX] = -7.0
        if base >= 8:
            L[4 * k:5 * k, abi.BX] = 0.0                  # Bx = 0
            R[4 * k:5 * k, abi.BX] = 0.0
            L[5 * k:6 * k, abi.BY:abi.BZ + 1] = 0.0       # B_t = 0
            R[5 * k:6 * k, abi.BY:abi.BZ + 1] = 0.0
        R[6 * k:7 * k, abi.VX] = L[6 * k:7 * k, abi.VX] + 6.0   # strong rarefaction
        R[7 * k:8 * k, abi.VX] = L[7 * k:8 * k, abi.VX] - 6.0   # strong compression
    return L, R
