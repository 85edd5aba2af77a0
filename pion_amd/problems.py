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
    return cfg, fill_hd_blast_octant(cfg, nzones)


def fill_hd_blast_octant(cfg, nzones):
    """IC of hd_blast_octant for any (global or slab) configuration, built plane-wise."""
    rho0, p0 = 2.338e-24, 1.38e-13
    P = alloc(cfg)
    x, y, z = cell_centres(cfg)
    rb = nzones * cfg.dx
    vol = 4.0 / 3.0 * math.pi * rb ** 3        # 1e51 erg in the full sphere
    pin = 1.0e51 * (cfg.gamma - 1.0) / vol
    P[abi.RO] = rho0
    xy2 = x[None, :] ** 2 + (y[:, None] ** 2 if cfg.ndim > 1 else 0.0)
    for k in range(z.size):
        r2 = xy2 + (z[k] ** 2 if cfg.ndim > 2 else 0.0)
        hot = r2 < rb * rb
        P[abi.PG, k] = np.where(hot, pin, p0)
        for t in range(cfg.ntracer):
            P[5 + t, k] = np.where(hot, 1.0, 0.0)
    return P


def mhd_blast_generic(ng, eqntype=abi.EQGLM, solver=abi.FLUX_RS_HLLD, strict_fp=0, ntracer=0):
    """The Stone blast of mhd_blastwave on an ng[0] x ng[1] x ng[2] box (dx = 1/ng[0], centred, periodic)
    without its degeneracies: B_z != 0 and a velocity field whose divergence has a definite sign at the
    blast edge.  On the symmetric blast (v = 0, B_z = 0 exactly) the reference itself is discontinuous:
    the HLLD -> HLL switch tests div v < 0 where div v is rounding noise, and for B_n -> +-0 the U** states
    flip with the sign of B_n (tests/test_reference_conditioning.py shows 1 ulp -> 2e-4 after two steps),
    so cell-wise fast-vs-oracle comparisons are made on this well-conditioned variant (1 ulp -> 1e-15)."""
    ng = list(ng)
    ndim = len(ng)
    dx = 1.0 / ng[0]
    xmin = tuple(-0.5 * ng[a] * dx if a < ndim else 0.0 for a in range(3))
    nvb = {abi.EQMHD: 8, abi.EQGLM: 9}[eqntype]
    cfg = abi.make_config(ndim, ng, eqntype, solver, ntracer=ntracer, artvisc=abi.AV_FKJ98_1D, etav=0.1,
                          gamma=5.0 / 3.0, cfl=0.24, dx=dx, xmin=xmin, bcs=["periodic"] * (2 * ndim),
                          refvec=[1.0, 0.1, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0][:nvb] + [1.0] * ntracer,
                          strict_fp=strict_fp)
    P = alloc(cfg)
    X, Y, Z = mesh(cfg)
    tp = 2.0 * math.pi
    Lx, Ly, Lz = ng[0] * dx, ng[1] * dx, (ng[2] * dx if ndim > 2 else 1.0)
    r2 = X * X + Y * Y + (Z * Z if ndim > 2 else 0.0)
    rb = min(0.1, 0.3 * min(Lx, Ly, Lz))
    hot = r2 < rb * rb
    P[abi.RO] = 1.0
    P[abi.PG] = np.where(hot, 10.0, 0.1)
    P[abi.VX] = 0.3 * np.sin(tp * X / Lx) + 0.05
    P[abi.VY] = 0.3 * np.sin(tp * Y / Ly) - 0.07
    P[abi.VZ] = (0.3 * np.sin(tp * Z / Lz) if ndim > 2 else 0.0) + 0.02
    P[abi.BX] = 1.0 / math.sqrt(2.0)
    P[abi.BY] = 1.0 / math.sqrt(2.0)
    P[abi.BZ] = 0.3 + 0.05 * np.sin(tp * X / Lx)
    for t in range(ntracer):
        P[nvb + t] = np.where(hot, 1.0 - 0.25 * t, 0.1 * t)
    return cfg, P


def hd_blast_box(ng, solver=abi.FLUX_RSroe, ntracer=0, artvisc=abi.AV_FKJ98_1D, nzones=3.0, strict_fp=0):
    """hd_blast_octant on an ng[0] x ng[1] x ng[2] box (same cell size on every axis)."""
    ng = list(ng)
    ndim = len(ng)
    L = 3.086e18
    dx = L / ng[0]
    bcs = []
    for a in range(ndim):
        bcs += ["reflecting", "outflow"]
    rho0, p0 = 2.338e-24, 1.38e-13
    cfg = abi.make_config(ndim, ng, abi.EQEUL, solver, ntracer=ntracer, artvisc=artvisc, etav=0.1,
                          gamma=5.0 / 3.0, cfl=0.3, dx=dx, xmin=(0.0, 0.0, 0.0), bcs=bcs,
                          refvec=[rho0, p0, 1e6, 1e6, 1e6] + [1.0] * ntracer, strict_fp=strict_fp)
    P = fill_hd_blast_octant(cfg, nzones)
    # smooth structure everywhere, so that every x-tile seam of the stage kernel carries gradients
    X, Y, Z = mesh(cfg)
    tp = 2.0 * math.pi / L
    P[abi.RO] *= 1.0 + 0.2 * np.sin(3 * tp * X) * np.cos(2 * tp * Y)
    P[abi.VX] = 1.0e5 * np.sin(2 * tp * X + 0.3)
    P[abi.VY] = 0.7e5 * np.cos(3 * tp * Y)
    if ndim > 2:
        P[abi.VZ] = -0.5e5 * np.sin(5 * tp * Z + 0.1)
    for t in range(ntracer):
        P[5 + t] = np.clip(P[5 + t] + 0.3 + 0.3 * np.sin(4 * tp * X), 0.0, 1.0)
    return cfg, P


def blast_axi2d(n, eqntype=abi.EQEUL, solver=abi.FLUX_RSroe, ntracer=0, artvisc=abi.AV_FKJ98_1D, strict_fp=0):
    """2-D axisymmetric (z,R) blast (test_problems/blastwave_axi2d): n x n/2 cells on z in [-1/2,1/2],
    R in [0,1/2]; the symmetry axis is the YN face (axisymmetric BC), outflow elsewhere.  MHD: uniform
    field along z plus a weak toroidal component so that every geometric source term is exercised."""
    ng = [n, n // 2]
    nvb = {abi.EQEUL: 5, abi.EQMHD: 8, abi.EQGLM: 9}[eqntype]
    ref = [1.0, 0.1, 1.0, 1.0, 1.0] + ([1.0, 1.0, 1.0] if nvb >= 8 else []) + ([1.0] if nvb == 9 else []) \
        + [1.0] * ntracer
    cfg = abi.make_config(2, ng, eqntype, solver, ntracer=ntracer, artvisc=artvisc, etav=0.1, gamma=5.0 / 3.0,
                          cfl=0.3, xmin=(-0.5, 0.0, 0.0), xmax=(0.5, 0.5, 0.0),
                          bcs=["outflow", "outflow", "axisymmetric", "outflow"], refvec=ref, strict_fp=strict_fp,
                          coord_sys=2)
    P = alloc(cfg)
    X, Y, Z = mesh(cfg)
    r2 = X * X + Y * Y
    P[abi.RO] = 1.0 + 0.3 * np.exp(-r2 / 0.05)
    P[abi.PG] = np.where(r2 < 0.15 * 0.15, 10.0, 0.1)
    P[abi.VX] = 0.2 * np.sin(2 * np.pi * X)
    P[abi.VY] = 0.1 * Y
    P[abi.VZ] = 0.3 * Y                      # rotation about the axis (v_theta)
    if nvb >= 8:
        P[abi.BX] = 0.5                      # along the axis
        P[abi.BY] = 0.05 * Y                 # radial, vanishing on the axis
        P[abi.BZ] = 0.2 * Y                  # toroidal
    if nvb == 9:
        P[abi.SI] = 0.01 * np.sin(2 * np.pi * X) * Y
    for t in range(ntracer):
        P[nvb + t] = np.where(r2 < 0.15 * 0.15, 1.0, 0.0)
    return cfg, P


def blast_sph1d(n, solver=abi.FLUX_RSroe, ntracer=0, artvisc=abi.AV_FKJ98_1D, strict_fp=0):
    """1-D spherically symmetric blast (test_problems/blastwave_sph1d): R in [0,1], reflecting at the
    origin, outflow outside; hydro only, as in the reference (sph_FV_solver_Hydro_Euler)."""
    cfg = abi.make_config(1, [n], abi.EQEUL, solver, ntracer=ntracer, artvisc=artvisc, etav=0.1, gamma=5.0 / 3.0,
                          cfl=0.3, xmin=(0.0, 0.0, 0.0), xmax=(1.0, 0.0, 0.0), bcs=["reflecting", "outflow"],
                          refvec=[1.0, 0.1, 1.0, 1.0, 1.0] + [1.0] * ntracer, strict_fp=strict_fp, coord_sys=3)
    P = alloc(cfg)
    X, Y, Z = mesh(cfg)
    P[abi.RO] = 1.0 + 0.5 * np.exp(-(X / 0.3) ** 2)
    P[abi.PG] = np.where(X < 0.2, 10.0, 0.1)
    P[abi.VX] = 0.3 * X
    for t in range(ntracer):
        P[5 + t] = np.where(X < 0.2, 1.0, 0.0)
    return cfg, P


def jet3d(n, solver=abi.FLUX_RSroe, jetradius=3, strict_fp=0):
    """3-D Cartesian hydro jet (ics/jet.cpp + boundaries/jet_boundaries.cpp): a uniform ambient medium,
    outflow on every face, and the internal JETBC on the XN face: a light transonic beam of `jetradius`
    cells centred on the x axis, carrying tracer 1.  Returns (cfg, P, (jetradius, jetstate)); the
    backend gets sim.set_jet(jetradius, jetstate) before init."""
    cfg = abi.make_config(3, [n, n, n], abi.EQEUL, solver, ntracer=1, artvisc=abi.AV_FKJ98_1D, etav=0.15,
                          gamma=5.0 / 3.0, cfl=0.3, xmin=(0.0, -0.5, -0.5), xmax=(1.0, 0.5, 0.5),
                          bcs=["outflow"] * 6, refvec=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0], strict_fp=strict_fp)
    P = alloc(cfg)
    P[abi.RO] = 1.0
    P[abi.PG] = 1.0
    jetstate = np.array([0.5, 1.0, 2.0, 0.0, 0.0, 1.0])   # mildly supersonic: stable from step one
    # (faster jets need the first-step limit of calc_timestep.cpp:313-323, SimControl.first_step_dt_limit)
    return cfg, P, (jetradius, jetstate)


def jet_axi2d(n, eqntype=abi.EQGLM, solver=abi.FLUX_RS_HLLD, jetradius=4, strict_fp=0, xn="outflow"):
    """2-D axisymmetric (z,R) magnetised jet (ics/jet.cpp, boundaries/jet_boundaries.cpp 2-D branch): uniform
    ambient medium with an axial field, axisymmetric BC on the axis, outflow elsewhere, internal JETBC on XN."""
    nvb = {abi.EQEUL: 5, abi.EQMHD: 8, abi.EQGLM: 9}[eqntype]
    ref = [1.0, 1.0, 1.0, 1.0, 1.0] + ([1.0, 1.0, 1.0] if nvb >= 8 else []) + ([1.0] if nvb == 9 else []) + [1.0]
    cfg = abi.make_config(2, [n, n // 2], eqntype, solver, ntracer=1, artvisc=abi.AV_FKJ98_1D, etav=0.15,
                          gamma=5.0 / 3.0, cfl=0.3, xmin=(0.0, 0.0, 0.0), xmax=(1.0, 0.5, 0.0),
                          bcs=[xn, "outflow", "axisymmetric", "outflow"], refvec=ref, strict_fp=strict_fp,
                          coord_sys=2)
    P = alloc(cfg)
    X, Y, Z = mesh(cfg)
    P[abi.RO] = 1.0
    P[abi.PG] = 1.0 + 2.0 * np.exp(-((X - 0.2) ** 2 + Y * Y) / 0.01)   # a pressure bump near the XN wall
    js = np.zeros(cfg.nvar)
    js[abi.RO], js[abi.PG], js[abi.VX] = 0.5, 1.0, 2.0
    if nvb >= 8:
        P[abi.BX] = 0.3
        js[abi.BX], js[abi.BY] = 0.3, 0.2   # axial field, toroidal field (JP.jetstate[BY] -> B_theta)
    js[nvb] = 1.0
    return cfg, P, (jetradius, js)


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


MSUN_PER_YR = 1.989e33 / 3.156e7   # g/s per Msun/yr (pconst.Msun()/pconst.year())


def wind_cells(cfg, pos, radius, mdot_msun_yr, vinf_kms, Tw, Rstar, tracers):
    """Stellar-wind boundary cells and their fixed states for a 3-D Cartesian grid:
    membership = every cell (ghosts included) whose centre is within `radius` of the source
    (boundaries/stellar_wind_boundaries.cpp:200-240); state = stellar_wind::
    set_wind_cell_reference_state (grid/stellar_wind_BC.cpp:375-470): rho = Mdot/(4 pi r^2 v_inf),
    adiabatic p from the surface temperature, radial velocity v_inf; cells inside 0.75 radius
    carry rho = p = 1e-31.  Returns (cell ids, states[n, nvar])."""
    kB, m_p = 1.38064852e-16, 1.672621898e-24
    mdot = mdot_msun_yr * MSUN_PER_YR
    vinf = vinf_kms * 1.0e5
    x, y, z = cell_centres(cfg)
    Z, Y, X = np.meshgrid(z - pos[2], y - pos[1], x - pos[0], indexing="ij")
    dist = np.sqrt(X * X + Y * Y + Z * Z)
    sel = dist <= radius
    idx = np.flatnonzero(sel.reshape(-1))
    d = dist.reshape(-1)[idx]
    st = np.zeros((idx.size, cfg.nvar))
    rho = 1.0 / d
    rho = rho * rho
    rho = rho * (mdot / (vinf * 4.0 * math.pi))
    pg = kB * Tw / m_p
    pg = pg * math.exp((cfg.gamma - 1.0) * math.log(4.0 * math.pi * Rstar * Rstar * vinf / mdot))
    pg = pg * np.exp(cfg.gamma * np.log(rho))
    inner = d < 0.75 * radius
    st[:, abi.RO] = np.where(inner, 1.0e-31, rho)
    st[:, abi.PG] = np.where(inner, 1.0e-31, pg)
    st[:, abi.VX] = vinf * X.reshape(-1)[idx] / d
    st[:, abi.VY] = vinf * Y.reshape(-1)[idx] / d
    st[:, abi.VZ] = vinf * Z.reshape(-1)[idx] / d
    base = cfg.nvar - cfg.ntracer
    for t in range(cfg.ntracer):
        st[:, base + t] = tracers[t]
    return idx.astype(np.int64), st


def wind3d(n, strict_fp=0):
    """M3: test_problems/Wind3D/params_Wind3D_n0128_l2.txt on a single level with n^3 cells:
    Euler + 1 tracer, FVS, FKJ98 eta 0.15, cooling 8 (mp_only_cooling), T in [5e3, 1e8],
    reflecting / one-way-outflow, a stellar wind at the origin.
    Returns (cfg, P, wind=(idx, states), first_step_dt_limit)."""
    L = 3.160064e18
    bcs = ["reflecting", "one-way-outflow"] * 3
    cfg = abi.make_config(3, [n, n, n], abi.EQEUL, abi.FLUX_FVS, ntracer=1, artvisc=abi.AV_FKJ98_1D, etav=0.15,
                          gamma=1.6666666666666667, cfl=0.3, xmin=(0.0, 0.0, 0.0), xmax=(L, L, L), bcs=bcs,
                          refvec=[1.0e-24, 1.0e-13, 1.0e6, 1.0e6, 1.0e6, 1.0], min_temp=5.0e3, max_temp=1.0e8,
                          cooling=abi.COOL_WSS09_CIE_LINE_HEAT_COOL, mp_timestep_limit=1, strict_fp=strict_fp)
    P, (idx, st), dt_lim = fill_wind3d(cfg, n)
    return cfg, P, (idx, st), dt_lim


def fill_wind3d(cfg, n_global):
    """IC, wind cells and first-step dt limit of wind3d for any (global or slab) configuration"""
    P = alloc(cfg)
    P[abi.RO] = 2.124229813e-24
    P[abi.PG] = 2.209037632e-12
    # keep the wind region resolved like the shipped set-up (12 cells at 128^3 on the finest of 2 levels)
    n = n_global
    radius = 1.543e17 * max(1.0, 256.0 / n) if n < 256 else 1.543e17
    idx, st = wind_cells(cfg, (0.0, 0.0, 0.0), radius, 1.0e-7, 1500.0, 3.0e4, 6.96e11, [1.0])
    for v in range(cfg.nvar):
        P[v].reshape(-1)[idx] = st[:, v]
    dt_lim = 0.1 * cfg.cfl * cfg.dx / (1500.0 * 1.0e5)   # calc_timestep.cpp:319-323
    return P, (idx, st), dt_lim


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


def cooling_blast3d(n, strict_fp=0, solver=abi.FLUX_FVS):
    """Wind3D's equations, solver, microphysics and units (test_problems/Wind3D/params_Wind3D_n0128_l2.txt:
    Euler + 1 tracer, FVS, FKJ98 eta 0.15, cooling 8 with the cooling-time step limit, T in [5e3, 1e8],
    reflecting / one-way-outflow) WITHOUT the stellar-wind source: a hot dense bubble (2e6 K, 20x the ambient
    density, tracer 1) at the origin of the photoionised ambient medium, so that both the radiative losses of
    the bubble and the heating/cooling balance of the ambient gas act from the first step.  This is the
    cooling configuration the reference objects can run end to end (the wind source needs GSL there)."""
    L = 3.160064e18
    bcs = ["reflecting", "one-way-outflow"] * 3
    cfg = abi.make_config(3, [n, n, n], abi.EQEUL, solver, ntracer=1, artvisc=abi.AV_FKJ98_1D, etav=0.15,
                          gamma=1.6666666666666667, cfl=0.3, xmin=(0.0, 0.0, 0.0), xmax=(L, L, L), bcs=bcs,
                          refvec=[1.0e-24, 1.0e-13, 1.0e6, 1.0e6, 1.0e6, 1.0], min_temp=5.0e3, max_temp=1.0e8,
                          cooling=abi.COOL_WSS09_CIE_LINE_HEAT_COOL, mp_timestep_limit=1, strict_fp=strict_fp)
    P = alloc(cfg)
    X, Y, Z = mesh(cfg)
    r2 = X * X + Y * Y + Z * Z
    inside = r2 < (0.3 * L) ** 2
    mu_over_kb = 0.609 * 1.672621898e-24 / 1.38064852e-16
    P[abi.RO] = np.where(inside, 20.0, 1.0) * 2.124229813e-24
    P[abi.PG] = P[abi.RO] * np.where(inside, 2.0e6, 7.5e3) / mu_over_kb
    P[5] = np.where(inside, 1.0, 0.0)
    return cfg, P
