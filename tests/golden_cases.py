"""Definitions of the golden-fixture cases, shared by tests/golden/make_golden.py (which runs the
reference) and the tests (which run the oracle / the GPU on the same inputs)."""
import numpy as np

from pion_amd import abi, problems

NFLUX = 160
GLM_DT = 0.01
NSTEPS = 2


def flux_cases():
    cases = []
    for sv in (0, 1, 2, 3, 4, 5, 6, 8):
        for ntr in (0, 2):
            for av in ((0, 1, 4) if sv == 4 else (1,)):
                cases.append((abi.EQEUL, sv, ntr, av))
    for eq in (abi.EQMHD, abi.EQGLM):
        for sv in (0, 7, 8):
            for ntr in (0, 1):
                for av in (0, 1):
                    cases.append((eq, sv, ntr, av))
    return cases


def flux_cases_b():
    """second fixture file (flux_kat_b.npz, own seed so that the first file's vectors never move):
    Roe-MHD (solver 4, with and without the H-correction eta) and the FKJ98 linear MHD solver (1)"""
    cases = []
    for eq in (abi.EQMHD, abi.EQGLM):
        for ntr in (0, 1):
            for av in (0, 1, 4):
                cases.append((eq, abi.FLUX_RSroe, ntr, av))
            for av in (0, 1):
                cases.append((eq, abi.FLUX_RSlinear, ntr, av))
    return cases


def flux_cfg(eq, sv, ntr, av, strict_fp=1):
    return abi.make_config(3, [4, 4, 4], eq, sv, ntracer=ntr, artvisc=av, xmax=(1, 1, 1), strict_fp=strict_fp)


def flux_key(eq, sv, ntr, av):
    return "eq%d_s%d_t%d_av%d" % (eq, sv, ntr, av)


def cell_cases():
    return [(abi.EQEUL, 0, 0), (abi.EQEUL, 1, 0), (abi.EQEUL, 1, 8), (abi.EQMHD, 0, 0), (abi.EQGLM, 0, 0),
            (abi.EQGLM, 1, 8)]


def cell_cfg(eq, ntr, cool, strict_fp=1):
    solver = abi.FLUX_RSroe if eq == abi.EQEUL else abi.FLUX_RS_HLLD
    return abi.make_config(3, [4, 4, 4], eq, solver, ntracer=ntr, xmax=(1, 1, 1), cooling=cool,
                           min_temp=1e-2, max_temp=1e9, strict_fp=strict_fp)


def cell_key(eq, ntr, cool):
    return "eq%d_t%d_c%d" % (eq, ntr, cool)


def cell_inputs(rng, cfg, n=200):
    L, _ = problems.random_states(rng, n, cfg.eqntype, cfg.ntracer)
    # make the states CGS-like when a microphysics object is present so that temperatures are sane
    if cfg.cooling:
        L[:, abi.RO] *= 1e-24
        L[:, abi.PG] *= 1e-12
    dU = np.zeros_like(L)
    scale = np.abs(L).max(axis=0)
    dU[:] = rng.normal(0, 0.05, L.shape) * scale
    dU[:, abi.RHO] = np.abs(dU[:, abi.RHO]) * 0.1
    # a third of the cells lose enough energy to need the negative-pressure repair
    k = n // 3
    dU[:k, abi.ERG] = -3.0 * (L[:k, abi.PG] / (cfg.gamma - 1.0) + 0.5 * L[:k, abi.RO] * (L[:k, 2:5] ** 2).sum(axis=1))
    return L, dU


STEP_CASES = ["hd_roe_3d", "hd_fvs_2d_tr", "hd_roe_hcorr_2d", "hd_hll_1d", "mhd_hlld_2d", "glm_hlld_3d",
              "glm_mixed_2d", "dmr_2d", "hd_lf_oa1_3d"]


STEP_CASES_B = ["mhd_roe_hcorr_2d", "glm_roe_3d", "glm_linear_2d", "hd_jet_3d",
                "cyl_hd_roe", "cyl_hd_hcorr_tr", "cyl_mhd_hlld", "cyl_glm_hlld", "cyl_glm_roe_oa1",
                "sph_hd_roe_tr", "sph_hd_hcorr", "sph_hd_hybrid_oa1", "cyl_glm_jet", "cyl_glm_jetreflect"]   # steps_b.npz (added with flux_kat_b.npz)


def step_setup(name, sim):
    """backend calls a case needs before init (besides the configuration)"""
    if name == "hd_jet_3d":
        _, _, (radius, state) = problems.jet3d(12)
        sim.set_jet(radius, state)
    if name in ("cyl_glm_jet", "cyl_glm_jetreflect"):
        _, _, (radius, state) = problems.jet_axi2d(24)
        sim.set_jet(radius, state)


def step_case(name, strict_fp=1):
    if name == "hd_jet_3d":
        cfg, P, _ = problems.jet3d(12, strict_fp=strict_fp)
        return cfg, P
    if name == "sph_hd_roe_tr":
        return problems.blast_sph1d(64, abi.FLUX_RSroe, ntracer=1, strict_fp=strict_fp)
    if name == "sph_hd_hcorr":
        return problems.blast_sph1d(64, abi.FLUX_RSroe, artvisc=abi.AV_HCORR_FKJ98, strict_fp=strict_fp)
    if name == "sph_hd_hybrid_oa1":
        cfg, P = problems.blast_sph1d(64, abi.FLUX_RShybrid, strict_fp=strict_fp)
        cfg.sp_ooa = cfg.tm_ooa = 1
        return cfg, P
    if name == "cyl_glm_jet":
        cfg, P, _ = problems.jet_axi2d(24, strict_fp=strict_fp)
        return cfg, P
    if name == "cyl_glm_jetreflect":
        cfg, P, _ = problems.jet_axi2d(24, strict_fp=strict_fp, xn="jetreflect")
        return cfg, P
    if name == "cyl_hd_roe":
        return problems.blast_axi2d(24, abi.EQEUL, abi.FLUX_RSroe, strict_fp=strict_fp)
    if name == "cyl_hd_hcorr_tr":
        return problems.blast_axi2d(24, abi.EQEUL, abi.FLUX_RSroe, ntracer=1, artvisc=abi.AV_HCORR_FKJ98,
                                    strict_fp=strict_fp)
    if name == "cyl_mhd_hlld":
        return problems.blast_axi2d(24, abi.EQMHD, abi.FLUX_RS_HLLD, strict_fp=strict_fp)
    if name == "cyl_glm_hlld":
        return problems.blast_axi2d(24, abi.EQGLM, abi.FLUX_RS_HLLD, ntracer=1, strict_fp=strict_fp)
    if name == "cyl_glm_roe_oa1":
        cfg, P = problems.blast_axi2d(24, abi.EQGLM, abi.FLUX_RSroe, strict_fp=strict_fp)
        cfg.sp_ooa = cfg.tm_ooa = 1
        return cfg, P
    if name == "mhd_roe_hcorr_2d":
        cfg, P = problems.mhd_blastwave(24, 2, abi.EQMHD, abi.FLUX_RSroe, strict_fp=strict_fp)
        cfg.artvisc = abi.AV_HCORR_FKJ98
        return cfg, P
    if name == "glm_roe_3d":
        return problems.mhd_blastwave(12, 3, abi.EQGLM, abi.FLUX_RSroe, strict_fp=strict_fp)
    if name == "glm_linear_2d":
        return problems.mhd_smooth(20, 2, abi.EQGLM, abi.FLUX_RSlinear, strict_fp=strict_fp,
                                   bcs=["outflow", "one-way-outflow", "reflecting", "outflow"])
    if name == "hd_roe_3d":
        return problems.hd_blast_octant(12, 3, solver=abi.FLUX_RSroe, strict_fp=strict_fp, nzones=3.0)
    if name == "hd_fvs_2d_tr":
        return problems.hd_blast_octant(24, 2, solver=abi.FLUX_FVS, ntracer=1, strict_fp=strict_fp, nzones=3.0)
    if name == "hd_roe_hcorr_2d":
        return problems.hd_blast_octant(24, 2, solver=abi.FLUX_RSroe, artvisc=abi.AV_HCORR_FKJ98,
                                        strict_fp=strict_fp, nzones=3.0)
    if name == "hd_hll_1d":
        return problems.hd_blast_octant(64, 1, solver=abi.FLUX_RS_HLL, strict_fp=strict_fp, nzones=3.0)
    if name == "mhd_hlld_2d":
        return problems.mhd_blastwave(24, 2, abi.EQMHD, abi.FLUX_RS_HLLD, strict_fp=strict_fp)
    if name == "glm_hlld_3d":
        return problems.mhd_blastwave(12, 3, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=strict_fp)
    if name == "glm_mixed_2d":
        return problems.mhd_smooth(20, 2, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=strict_fp,
                                   bcs=["outflow", "one-way-outflow", "reflecting", "outflow"])
    if name == "dmr_2d":
        return problems.double_mach_reflection(52, strict_fp=strict_fp)
    if name == "hd_lf_oa1_3d":
        cfg, P = problems.hd_blast_octant(10, 3, solver=abi.FLUX_LF, strict_fp=strict_fp, nzones=3.0)
        cfg.sp_ooa = cfg.tm_ooa = 1
        return cfg, P
    raise KeyError(name)


# ---- long runs: low-resolution versions of the BASELINE configurations, end state from the reference ----
# (endstate.npz, made by `make_golden.py c`).  Parameter values are the shipped parameter files' (cited per
# case); the initial data are our own closed forms of the same set-ups (no sub-cell blending), the same
# arrays being fed to the reference objects, the oracle and the GPU.  Config 5 (Wind3D + cooling) is not
# here: its cooling tables are "parity unpinned" (no GSL in the build container), see DESIGN.md s2.
END_CASES = ["sph1d_n128", "dmr_n065", "mhd_bw2d_64x96", "mhd_ideal_bw2d_64x96", "bw3d_nr032"]


def end_case(name, strict_fp=1):
    """-> (cfg, P, finishtime, max_steps)"""
    L = 30.86e18
    if name == "sph1d_n128":
        # test_problems/blastwave_sph1d/params_sphBW_n128.txt: spherical 1-D Euler, hybrid solver (3),
        # 1e51 erg in BW_nzones = 1 cell, ambient 2.34e-22 / 1.38e-11, reflecting / outflow, CFL 0.3
        n = 128
        cfg = abi.make_config(1, [n], abi.EQEUL, abi.FLUX_RShybrid, artvisc=abi.AV_FKJ98_1D, etav=0.1, gamma=5.0 / 3.0,
                              cfl=0.3, xmin=(0.0, 0.0, 0.0), xmax=(L, 0.0, 0.0), bcs=["reflecting", "outflow"],
                              refvec=[1.0e-23, 3.0e-10, 1.0e6, 1.0e6, 1.0e6], strict_fp=strict_fp, coord_sys=3)
        P = problems.alloc(cfg)
        X, _, _ = problems.mesh(cfg)
        rb = 1.0 * cfg.dx
        P[abi.RO] = 2.34e-22
        P[abi.PG] = np.where(X < rb, 1.0e51 * (cfg.gamma - 1.0) / (4.0 / 3.0 * np.pi * rb ** 3), 1.38e-11)
        return cfg, P, 1.58e12, 400
    if name == "dmr_n065":
        # test_problems/double_Mach_reflection/params_DMR_n065.txt: 65 x 20, Roe-CV (4), gamma 1.4, CFL 0.4, t = 0.2
        cfg, P = problems.double_mach_reflection(65, strict_fp=strict_fp)
        return cfg, P, 0.2, 2000
    if name in ("mhd_bw2d_64x96", "mhd_ideal_bw2d_64x96"):
        # test_problems/MHD_Blastwave2D/params_MHD_blastwave2D_UG_B010_n256.txt at 64 x 96: [-1/2,1/2] x
        # [-3/4,3/4], periodic, HLLD (7), CFL 0.24, eta 0.1, t = 0.2; glm-mhd as shipped, and ideal MHD
        eq = abi.EQGLM if name == "mhd_bw2d_64x96" else abi.EQMHD
        cfg = abi.make_config(2, [64, 96], eq, abi.FLUX_RS_HLLD, artvisc=abi.AV_FKJ98_1D, etav=0.1, gamma=5.0 / 3.0,
                              cfl=0.24, dx=1.0 / 64, xmin=(-0.5, -0.75, 0.0), bcs=["periodic"] * 4,
                              refvec=[1.0, 0.1, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0], strict_fp=strict_fp)
        P = problems.alloc(cfg)
        X, Y, _ = problems.mesh(cfg)
        P[abi.RO] = 1.0
        P[abi.PG] = np.where(X * X + Y * Y < 0.01, 10.0, 0.1)
        P[abi.BX] = 1.0 / np.sqrt(2.0)
        P[abi.BY] = 1.0 / np.sqrt(2.0)
        return cfg, P, 0.2, 2000
    if name == "bw3d_nr032":
        # test_problems/blastwave_crt3d/params_BWcrt3D_Octant_NR032.txt: 32^3 octant, Roe-CV (4), 1e51 erg in
        # BW_nzones = 2 cells, ambient 2.34e-22 / 1.38e-11, reflecting / outflow, CFL 0.3; the first 80 steps
        bcs = ["reflecting", "outflow"] * 3
        cfg = abi.make_config(3, [32, 32, 32], abi.EQEUL, abi.FLUX_RSroe, artvisc=abi.AV_FKJ98_1D, etav=0.1,
                              gamma=5.0 / 3.0, cfl=0.3, xmin=(0.0, 0.0, 0.0), xmax=(L, L, L), bcs=bcs,
                              refvec=[1.0e-23, 3.0e-10, 1.0e6, 1.0e6, 1.0e6], strict_fp=strict_fp)
        P = problems.alloc(cfg)
        X, Y, Z = problems.mesh(cfg)
        rb = 2.0 * cfg.dx
        P[abi.RO] = 2.34e-22
        P[abi.PG] = np.where(X * X + Y * Y + Z * Z < rb * rb, 1.0e51 * (cfg.gamma - 1.0) / (4.0 / 3.0 * np.pi * rb ** 3),
                             1.38e-11)
        return cfg, P, 1.58e12, 80
    raise KeyError(name)


def end_run(sim, cfg, P, finishtime, max_steps):
    """the time loop of sim_control::Time_Int on any backend -> (steps taken, end time, list of dt)"""
    from pion_amd import driver
    sc = driver.SimControl(sim, cfg, finishtime=finishtime)
    sc.init(P)
    dts = []
    while sc.simtime < finishtime and len(dts) < max_steps:
        dts.append(sc.calculate_timestep())
        sc.advance_time()
    sc.finish_halo()
    return len(dts), sc.simtime, np.array(dts)


def conserved_totals(cfg, A):
    """sums over the on-grid cells of rho, rho v, E (and B): the BASELINE conservation gate"""
    nb = cfg.nbc
    sl = [slice(None)] + [slice(nb, -nb) if a < cfg.ndim else slice(None) for a in (2, 1, 0)]
    q = A[tuple(sl)]
    ro, pg, v = q[0], q[1], q[2:5]
    E = pg / (cfg.gamma - 1.0) + 0.5 * ro * (v ** 2).sum(axis=0)
    vals = [ro, ro * v[0], ro * v[1], ro * v[2]]
    if cfg.eqntype != abi.EQEUL:
        B = q[5:8]
        E = E + 0.5 * (B ** 2).sum(axis=0)
        vals += [B[0], B[1], B[2]]
    vals.append(E)
    return np.array([x.sum() for x in vals]), np.array([np.abs(x).sum() for x in vals])


def on_grid(cfg, A):
    nb = cfg.nbc
    sl = [slice(None)] + [slice(nb, -nb) if a < cfg.ndim else slice(None) for a in (2, 1, 0)]
    return A[tuple(sl)]


def diff_norms(cfg, a, b):
    """per-variable L1 / L2 / max of a-b over the on-grid cells, in units of refvec
    (the norms of analysis/silocompare/silocompare.cpp:371-430)"""
    nv = cfg.nvar
    d = np.abs(on_grid(cfg, a) - on_grid(cfg, b)).reshape(nv, -1)
    rv = np.array([cfg.refvec[v] for v in range(nv)])
    return d.mean(axis=1) / rv, np.sqrt((d * d).mean(axis=1)) / rv, d.max(axis=1) / rv


# ---- cooling ODE known-answer test -------------------------------------------------------------
ODE_TMIN, ODE_TMAX, ODE_NT = 1.0e2, 1.0e8, 200
ODE_GAMMA = 5.0 / 3.0
ODE_MU_TOT_OVER_KB = 0.609 * 1.672621898e-24 / 1.38064852e-16
ODE_MU = 1.40 * 1.672621898e-24
ODE_RHO = ODE_MU          # rho^2 * inv_Mu2 = 1


def ode_cie(T):
    """a smooth positive 'cooling curve' with a peak, tabulated on the log-T grid"""
    lt = np.log10(T)
    return 1.0e-22 * (0.05 + np.exp(-0.5 * ((lt - 5.2) / 0.6) ** 2)) * (T / 1.0e5) ** 0.3 * 1e-2


def ode_grid():
    dlog = (np.log10(ODE_TMAX) - np.log10(ODE_TMIN)) / (ODE_NT - 1)
    return np.array([10.0 ** (np.log10(ODE_TMIN) + i * dlog) for i in range(ODE_NT)])


def ode_table():
    """(E grid, dE/dt) equivalent to the oracle's Edot with only C_cie set: dE/dt = -C_cie(T(E))."""
    T = ode_grid()
    E = T * ODE_RHO / ((ODE_GAMMA - 1.0) * ODE_MU_TOT_OVER_KB)
    return np.ascontiguousarray(E), np.ascontiguousarray(-ode_cie(T))


def ode_cooling_tables():
    T = ode_grid()
    tabs = np.zeros((5, ODE_NT))
    tabs[4] = ode_cie(T)
    slopes = np.zeros((5, ODE_NT))
    slopes[4, :-1] = (tabs[4, 1:] - tabs[4, :-1]) / (T[1:] - T[:-1])
    return T, tabs, slopes


def ode_inputs(rng, n=300):
    T0 = 10.0 ** rng.uniform(2.5, 7.5, n)
    E0 = T0 * ODE_RHO / ((ODE_GAMMA - 1.0) * ODE_MU_TOT_OVER_KB)
    tcool = E0 / ode_cie(T0)
    dts = tcool * 10.0 ** rng.uniform(-4, 0.7, n)
    return np.ascontiguousarray(E0), np.ascontiguousarray(dts)


# ---- cooling known-answer vectors against the REFERENCE's mp_only_cooling (cooling_kat.npz) -------------
# (min_temp, max_temp) of EP: Wind3D's range; a range whose limits mp_only_cooling.cpp:140-146 overrides
# (MinT_allowed 1e-2 -> 1, MaxT_allowed 5e10 -> 1e8, while the table still spans [1e-2, 5e10]); a cold range
COOL_RANGES = [(5.0e3, 1.0e8), (1.0e-2, 5.0e10), (1.0e1, 1.0e7)]
COOL_GAMMA = 5.0 / 3.0
COOL_NVAR = 6   # Euler + 1 tracer, as Wind3D


def cool_cfg(rng_idx, strict_fp=1):
    tmin, tmax = COOL_RANGES[rng_idx]
    return abi.make_config(3, [4, 4, 4], abi.EQEUL, abi.FLUX_FVS, ntracer=1, xmax=(1, 1, 1), gamma=COOL_GAMMA,
                           cooling=abi.COOL_WSS09_CIE_LINE_HEAT_COOL, min_temp=tmin, max_temp=tmax,
                           mp_timestep_limit=1, strict_fp=strict_fp)


def cool_edot_inputs(rng, tmin, tmax, Tgrid, n=1000):
    """(rho, T) pairs: log-uniform, half a decade beyond either end of the table, plus exact table nodes
    and their neighbours in floating point (the bisection's `<` at a node)"""
    rho = 10.0 ** rng.uniform(-26, -19, n)
    T = 10.0 ** rng.uniform(np.log10(tmin) - 0.5, np.log10(tmax) + 0.5, n)
    k = rng.integers(0, Tgrid.size, 60)
    T[:60] = Tgrid[k]
    T[60:80] = np.nextafter(Tgrid[k[:20]], 0.0)
    T[80:100] = np.nextafter(Tgrid[k[20:40]], np.inf)
    return rho, T


def cool_states(rng, tmin, tmax, n=500):
    """primitive states (Euler + tracer) at temperatures from a decade below MinTemperature to half a decade above
    MaxTemperature"""
    P = np.zeros((n, COOL_NVAR))
    P[:, abi.RO] = 10.0 ** rng.uniform(-25, -20, n)
    T = 10.0 ** rng.uniform(np.log10(max(tmin, 1.0)) - 1.0, np.log10(min(tmax, 1.0e8)) + 0.5, n)
    P[:, abi.PG] = P[:, abi.RO] * T / ODE_MU_TOT_OVER_KB
    P[:, abi.VX:abi.VZ + 1] = rng.normal(0, 1.0e6, (n, 3))
    P[:, 5] = rng.uniform(0, 1, n)
    return P, T


COOL_DTS = [1.0e3, 1.0e9, 3.0e10, 1.0e12]   # seconds: Euler shortcut ... many cooling times


STEP_CASES_C = ["cool_fvs_3d", "cool_roe_3d"]        # steps_c.npz: whole steps with the reference's mp_only_cooling
END_CASES_C = ["cool3d_n20", "mhd_ideal_generic_64x96"]   # endstate_c.npz


def step_case_c(name, strict_fp=1):
    if name == "cool_fvs_3d":
        return problems.cooling_blast3d(12, strict_fp=strict_fp)
    if name == "cool_roe_3d":
        return problems.cooling_blast3d(10, strict_fp=strict_fp, solver=abi.FLUX_RSroe)
    raise KeyError(name)


def end_case_c(name, strict_fp=1):
    if name == "cool3d_n20":
        cfg, P = problems.cooling_blast3d(20, strict_fp=strict_fp)
        return cfg, P, 1.0e30, 60
    if name == "mhd_ideal_generic_64x96":
        # ideal MHD + HLLD (the solver of BASELINE config 3 / the nvar-8 row of the headline) on a blast WITHOUT the
        # degeneracies of the shipped symmetric one (B_z != 0, a velocity field of definite divergence:
        # problems.mhd_blast_generic; the reference is continuous there, tests/test_reference_conditioning.py), 220
        # steps: the long-run cell-wise gate of the fast ideal-MHD path
        cfg, P = problems.mhd_blast_generic([64, 96], abi.EQMHD, abi.FLUX_RS_HLLD, strict_fp=strict_fp)
        return cfg, P, 1.0e30, 220
    raise KeyError(name)


# ---- the shock tubes the reference ships as initial conditions (shocktubes.npz, `make_golden.py e`) -------------
# Left / right states, interface position, gamma and finish time of IC_shocktube::get_riemann_ics
# (source/ics/shock_tube.cpp:473-815; the same table in dataIO/dataio_text.cpp:734-1080): Toro's tests 1-5
# (:477-530), Brio & Wu and Falle's FS / SS / FR / SR / OFS (:546-652), Ryu & Jones 1a-5b (:657-815; the reference
# sets no finish time for those: the times of Ryu & Jones 1995, ApJ 442, 228, figs 1-5 are used).  Run set-up of
# test_problems/untested/test_ShockTubes/pf_st_toro*.txt (Euler, hybrid solver 3, CFL 0.7, FKJ98 viscosity eta 0.3)
# and pf_st_falle*.txt (ideal MHD, HLLD, CFL 0.7, no artificial viscosity): 1-D, 200 cells on [0, 1], outflow.
_S4PI = 1.0 / np.sqrt(4.0 * np.pi)
# name: (gamma, xm, finishtime, left, right); states as (rho, p, vx, vy, vz[, Bx, By, Bz])
SHOCK_TUBES_HD = {
    "toro1": (1.4, 0.3, 0.2, (1.0, 1.0, 0.75, 0.0, 0.0), (0.125, 0.1, 0.0, 0.0, 0.0)),
    "toro2": (1.4, 0.5, 0.15, (1.0, 0.4, -2.0, 0.0, 0.0), (1.0, 0.4, 2.0, 0.0, 0.0)),
    "toro3": (1.4, 0.5, 0.012, (1.0, 1000.0, 0.0, 0.0, 0.0), (1.0, 0.01, 0.0, 0.0, 0.0)),
    "toro4": (1.4, 0.4, 0.035, (5.99924, 460.894, 19.5975, 0.0, 0.0), (5.99242, 46.0950, -6.19633, 0.0, 0.0)),
    "toro5": (1.4, 0.8, 0.012, (1.0, 1000.0, -19.59745, 0.0, 0.0), (1.0, 0.01, -19.59745, 0.0, 0.0)),
}
SHOCK_TUBES_MHD = {
    "falle_bw": (2.0, 0.5, 0.12, (1.0, 1.0, 0.0, 0.0, 0.0, 0.75, 1.0, 0.0), (0.125, 0.1, 0.0, 0.0, 0.0, 0.75, -1.0, 0.0)),
    "falle_fs": (5.0 / 3.0, 0.3, 0.4, (3.0, 16.33, -0.732, -1.333, 0.0, 3.0, 2.309, 0.0), (1.0, 1.0, -4.196, 0.0, 0.0, 3.0, 0.0, 0.0)),
    "falle_ss": (5.0 / 3.0, 0.3, 0.5, (1.368, 1.769, 0.269, 1.0, 0.0, 1.0, 0.0, 0.0), (1.0, 1.0, 0.0, 0.0, 0.0, 1.0, 1.0, 0.0)),
    "falle_fr": (5.0 / 3.0, 0.5, 0.1, (1.0, 2.0, 0.0, 0.0, 0.0, 1.0, 3.0, 0.0), (0.2641, 0.2175, 3.6, -2.551, 0.0, 1.0, 0.0, 0.0)),
    "falle_sr": (5.0 / 3.0, 0.5, 0.3, (1.0, 2.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0), (0.2, 0.1368, 1.186, 2.967, 0.0, 1.0, 1.6405, 0.0)),
    "falle_ofs": (5.0 / 3.0, 0.5, 0.15, (1.0, 1.0, 6.505, 1.0, 0.0, 1.0, 1.0, 1.0), (3.0, 20.268, 2.169, 1.331, 0.331, 1.0, 3.153, 3.153)),
    "rj1a": (5.0 / 3.0, 0.5, 0.08, (1.0, 20.0, 10.0, 0.0, 0.0, 5 * _S4PI, 5 * _S4PI, 0.0), (1.0, 1.0, -10.0, 0.0, 0.0, 5 * _S4PI, 5 * _S4PI, 0.0)),
    "rj1b": (5.0 / 3.0, 0.5, 0.03, (1.0, 1.0, 0.0, 0.0, 0.0, 3 * _S4PI, 5 * _S4PI, 0.0), (0.1, 10.0, 0.0, 0.0, 0.0, 3 * _S4PI, 2 * _S4PI, 0.0)),
    "rj2a": (5.0 / 3.0, 0.5, 0.2, (1.08, 0.95, 1.2, 0.01, 0.5, 2 * _S4PI, 3.6 * _S4PI, 2 * _S4PI), (1.0, 1.0, 0.0, 0.0, 0.0, 2 * _S4PI, 4 * _S4PI, 2 * _S4PI)),
    "rj2b": (5.0 / 3.0, 0.5, 0.035, (1.0, 1.0, 0.0, 0.0, 0.0, 3 * _S4PI, 6 * _S4PI, 0.0), (0.1, 10.0, 0.0, 2.0, 1.0, 3 * _S4PI, 1 * _S4PI, 0.0)),
    "rj3a": (5.0 / 3.0, 0.5, 0.01, (0.1, 0.4, 50.0, 0.0, 0.0, 0.0, -1 * _S4PI, -2 * _S4PI), (0.1, 0.2, 0.0, 0.0, 0.0, 0.0, 1 * _S4PI, 2 * _S4PI)),
    "rj3b": (5.0 / 3.0, 0.5, 0.1, (1.0, 1.0, -1.0, 0.0, 0.0, 0.0, 1.0, 0.0), (1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.0)),
    "rj4a": (5.0 / 3.0, 0.5, 0.15, (1.0, 1.0, 0.0, 0.0, 0.0, 1.0, 1.0, 0.0), (0.2, 0.1, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0)),
    "rj4b": (5.0 / 3.0, 0.5, 0.15, (0.4, 0.52467, -0.66991, 0.98263, 0.0, 1.3, 0.0025293, 0.0), (1.0, 1.0, 0.0, 0.0, 0.0, 1.3, 1.0, 0.0)),
    "rj4c": (5.0 / 3.0, 0.5, 0.15, (0.65, 0.5, 0.667, -0.257, 0.0, 0.75, 0.55, 0.0), (1.0, 0.75, 0.4, -0.94, 0.0, 0.75, 0.0, 0.0)),
    "rj4d": (5.0 / 3.0, 0.5, 0.16, (1.0, 1.0, 0.0, 0.0, 0.0, 0.7, 0.0, 0.0), (0.3, 0.2, 0.0, 0.0, 1.0, 0.7, 1.0, 0.0)),
    "rj5a": (5.0 / 3.0, 0.5, 0.1, (1.0, 1.0, 0.0, 0.0, 0.0, 0.75, 1.0, 0.0), (0.125, 0.1, 0.0, 0.0, 0.0, 0.75, -1.0, 0.0)),
    "rj5b": (5.0 / 3.0, 0.5, 0.16, (1.0, 1.0, 0.0, 0.0, 0.0, 1.3, 1.0, 0.0), (0.4, 0.4, 0.0, 0.0, 0.0, 1.3, -1.0, 0.0)),
}
SHOCK_TUBES = list(SHOCK_TUBES_HD) + list(SHOCK_TUBES_MHD)


def shock_tube_case(name, strict_fp=1, nx=200):
    """-> (cfg, P, finishtime)"""
    if name in SHOCK_TUBES_HD:
        gam, xm, tf, L, R = SHOCK_TUBES_HD[name]
        cfg = abi.make_config(1, [nx], abi.EQEUL, abi.FLUX_RShybrid, artvisc=abi.AV_FKJ98_1D, etav=0.3, gamma=gam,
                              cfl=0.7, xmin=(0.0, 0.0, 0.0), xmax=(1.0, 0.0, 0.0), bcs=["outflow", "outflow"],
                              refvec=[1.0] * 5, strict_fp=strict_fp)
    else:
        gam, xm, tf, L, R = SHOCK_TUBES_MHD[name]
        cfg = abi.make_config(1, [nx], abi.EQMHD, abi.FLUX_RS_HLLD, artvisc=0, etav=0.15, gamma=gam,
                              cfl=0.7, xmin=(0.0, 0.0, 0.0), xmax=(1.0, 0.0, 0.0), bcs=["outflow", "outflow"],
                              refvec=[1.0] * 8, strict_fp=strict_fp)
    P = problems.alloc(cfg)
    X, _, _ = problems.mesh(cfg)
    for v in range(cfg.nvar):
        P[v] = np.where(X < xm, L[v], R[v])
    return cfg, P, tf


# ---- more of the reference's shipped uniform-grid test problems (endstate_s.npz, `make_golden.py f`) ----------------
# Parameter files under /root/reference/test_problems/, initial conditions restated from source/ics/ (file:line in
# each case); run by the reference's solver objects to the shipped finish time or a step cap (CPU seconds).
END_CASES_S = ["fieldloop100", "fieldloop100_vz", "fieldloop100_static", "advection_cd3_n128", "lwi_n064",
               "oblique_m25_roe_fkj", "oblique_m25_roe_hcorr", "oblique_m25_fvs_fkj", "oblique_m40_roe_hcorr",
               "bwaxi2d_halfplane_nr016", "mhdbwaxi2d_halfplane_nr032"]


def _oblique(mach, solver, artvisc, strict_fp):
    # test_problems/ObliqueShock/params_oblique_shock_M25.txt / _M40.txt: 100 x 50 on [0,1e17] x [0,0.5e17], Euler +
    # one tracer, gamma 5/3, XN outflow / XP fixed / YN, YP outflow; run_ObliqueShockTest.sh runs each with cfl 0.4,
    # eta 0.1 and {Roe-CV + FKJ98, Roe-CV + H-correction, FVS + FKJ98}.  IC_shocktube::setup_data / assign_data
    # (ics/shock_tube.cpp:60-356): STnumber <= 0 reads pre / post-shock vectors; BOTH tracer parameters land in the
    # PRE-shock vector (:273-285), so the post-shock tracer stays 0 and the pre-shock one is STpostvecTR0 = -1;
    # the states are rotated by STangleXY = 2 degrees (eqns_base::rotateXY) and the interface is the line
    # x0(y) = shockpos + (0.5 - Ymin) tan(a) - (y - Ymin) tan(a), left (post-shock) state where x <= x0.
    post = {25: (3.9808917197e-22, 7.81e-10, -8.2074451397e05), 40: None}
    pre = {25: (1.0e-22, 1.0e-12, -3.237486122e6), 40: None}
    if mach == 40:
        # params_oblique_shock_M40.txt
        post[40] = (3.9925140362e-22, 1.99975e-09, -1.2934150634e+06)
        pre[40] = (1.0e-22, 1.0e-12, -5.1639777950e+06)
    cfg = abi.make_config(2, [100, 50], abi.EQEUL, solver, ntracer=1, artvisc=artvisc, etav=0.1, gamma=1.666666666666666666,
                          cfl=0.4, xmin=(0.0, 0.0, 0.0), xmax=(1.0e17, 0.5e17, 0.0),
                          bcs=["outflow", "fixed", "outflow", "outflow"],
                          refvec=[1.0e-22, 1.0e-10, 1.0e6, 1.0e6, 1.0e6, 1.0], strict_fp=strict_fp)
    ang = 2.0 * np.pi / 180.0
    ct, st, tt = np.cos(ang), np.sin(ang), np.tan(ang)

    def rot(s):
        ro, pg, vx = s
        return (ro, pg, vx * ct, vx * st, 0.0)   # rotateXY of (vx, 0, 0)
    left, right = rot(post[mach]) + (0.0,), rot(pre[mach]) + (-1.0,)
    P = problems.alloc(cfg)
    X, Y, _ = problems.mesh(cfg)
    x0 = (4.0e16 + 0.5 * tt) - Y * tt
    for v in range(cfg.nvar):
        P[v] = np.where(X <= x0, left[v], right[v])
    return cfg, P


# params_MHDaxi2dBW_HalfPlane_NR128.txt: glm-mhd with the Roe solver (4), CFL 0.2, FKJ98 eta 0.15, 1e51 erg in
# BW_nzones = 8 cells of the 256 x 128 grid (= 2 cells at 64 x 32), BWmagfieldX = 5.25357e-6 G along the axis
_MHDBW = {"eq": abi.EQGLM, "solver": abi.FLUX_RSroe, "av": abi.AV_FKJ98_1D, "eta": 0.15, "cfl": 0.2,
          "bcs": ["outflow", "outflow", "reflecting", "outflow"], "nzones": 2.0, "ro": 2.34e-22, "pg": 1.38e-11,
          "energy": 1.0e51, "tf": 1.578e12}
_MHDBW_BX = 5.25357e-06 / np.sqrt(4.0 * np.pi)


def end_case_s(name, strict_fp=1):
    """-> (cfg, P, finishtime, max_steps)"""
    if name.startswith("fieldloop100"):
        # test_problems/FieldLoop/params_FieldLoop100{,vz,Static}.txt: glm-mhd, HLLD (7), 100 x 50 on [-1,1] x [-1/2,1/2],
        # periodic, CFL 0.4, FKJ98 eta 0.1, t = 2.  IC_basic_tests::setup_FieldLoop (ics/basic_tests.cpp:553-665):
        # rho = p = 1, v = (2, 1, vz) (static: 0), A_z = 1e-3 (0.3 - r) inside r < 0.3, B = curl A by central
        # differences (VectorOps_Cart::Curl, coord_sys/VectorOps.cpp:446-528)
        vz = {"fieldloop100": 0.0, "fieldloop100_vz": 1.0, "fieldloop100_static": 0.0}[name]
        vel = 0.0 if name.endswith("static") else 2.0
        cfg = abi.make_config(2, [100, 50], abi.EQGLM, abi.FLUX_RS_HLLD, artvisc=abi.AV_FKJ98_1D, etav=0.1,
                              gamma=5.0 / 3.0, cfl=0.4, xmin=(-1.0, -0.5, 0.0), xmax=(1.0, 0.5, 0.0), bcs=["periodic"] * 4,
                              refvec=[1.0] * 9, strict_fp=strict_fp)
        P = problems.alloc(cfg)
        X, Y, _ = problems.mesh(cfg)
        P[abi.RO] = 1.0
        P[abi.PG] = 1.0
        P[abi.VX] = vel
        P[abi.VY] = vel / 2.0
        P[abi.VZ] = vz
        r = np.sqrt(X * X + Y * Y)
        A = np.where(r < 0.3, 0.001 * (0.3 - r), 0.0)
        # (the ghost cells hold no potential when the reference takes the curl; the loop does not reach them)
        nb = cfg.nbc
        A[:, :nb, :] = A[:, -nb:, :] = 0.0
        A[:, :, :nb] = A[:, :, -nb:] = 0.0
        Bx = np.zeros_like(A)
        By = np.zeros_like(A)
        Bx[:, 1:-1, :] = (A[:, 2:, :] - A[:, :-2, :]) / (2.0 * cfg.dx)
        By[:, :, 1:-1] = -(A[:, :, 2:] - A[:, :, :-2]) / (2.0 * cfg.dx)
        P[abi.BX], P[abi.BY] = Bx, By
        return cfg, P, 2.0, {"fieldloop100": 2000, "fieldloop100_vz": 120, "fieldloop100_static": 120}[name]
    if name == "advection_cd3_n128":
        # test_problems/advection/params_advection_v020t30_l1n128.txt: StarBench_ContactDiscontinuity3 (ics/
        # StarBench_test.cpp:156-300): Euler + one tracer, gamma 1.0001, Roe-CV (4), FKJ98 eta 0.15, CFL 0.4, 128^2 on
        # [0,2]^2, periodic; a square of rho = 10 rotated by 1 radian about (1,1), p = 10, v = (1.78884, 0.89443)
        cfg = abi.make_config(2, [128, 128], abi.EQEUL, abi.FLUX_RSroe, ntracer=1, artvisc=abi.AV_FKJ98_1D, etav=0.15,
                              gamma=1.0001, cfl=0.4, xmin=(0.0, 0.0, 0.0), xmax=(2.0, 2.0, 0.0), bcs=["periodic"] * 4,
                              refvec=[1.0] * 6, strict_fp=strict_fp)
        P = problems.alloc(cfg)
        X, Y, _ = problems.mesh(cfg)
        tt = np.tan(1.0)
        itt = 1.0 / tt
        ifst = 1.0 / (4.0 * np.sin(1.0))
        inside = ~((Y > 1.0 + itt + ifst - X * itt) | (Y < 1.0 + itt - ifst - X * itt)
                   | (Y > tt * (X - (1.0 - itt - ifst))) | (Y < tt * (X - (1.0 - itt + ifst))))
        P[abi.RO] = np.where(inside, 10.0, 1.0)
        P[abi.PG] = 10.0
        P[abi.VX] = 1.78884
        P[abi.VY] = 0.89443
        P[5] = np.where(inside, 1.0, 0.0)
        return cfg, P, 2.2360679775, 150
    if name == "lwi_n064":
        # test_problems/LiskaWendroffImplosion/params_LWI_d2l1n400.txt at 64^2: Euler, gamma 1.4, Roe-CV (4), NO
        # artificial viscosity, CFL 0.3, [0,0.3]^2, reflecting walls, t = 2.5; setup_LWImplosion
        # (ics/basic_tests.cpp:923-954): rho = p = 1, (0.125, 0.14) below the diagonal x + y < 0.15
        cfg = abi.make_config(2, [64, 64], abi.EQEUL, abi.FLUX_RSroe, artvisc=abi.AV_NONE, etav=0.15, gamma=1.4,
                              cfl=0.3, xmin=(0.0, 0.0, 0.0), xmax=(0.3, 0.3, 0.0), bcs=["reflecting"] * 4,
                              refvec=[1.0] * 5, strict_fp=strict_fp)
        P = problems.alloc(cfg)
        X, Y, _ = problems.mesh(cfg)
        low = (X < 0.15) & (Y < (0.15 - X))
        P[abi.RO] = np.where(low, 0.125, 1.0)
        P[abi.PG] = np.where(low, 0.14, 1.0)
        return cfg, P, 2.5, 400
    if name.startswith("oblique_"):
        mach = 25 if "_m25_" in name else 40
        solver = abi.FLUX_FVS if "_fvs_" in name else abi.FLUX_RSroe
        av = abi.AV_HCORRECTION if name.endswith("hcorr") else abi.AV_FKJ98_1D
        cfg, P = _oblique(mach, solver, av, strict_fp)
        return cfg, P, 9.48e10, 250
    if name == "bwaxi2d_halfplane_nr016":
        # test_problems/blastwave_axi2d/params_axi2dBW_HalfPlane_NR016.txt: cylindrical (z,R) 32 x 16, Euler, Roe-CV,
        # FKJ98 0.1, CFL 0.3, outflow / reflecting axis / outflow, 1e51 erg in BW_nzones = 2 cells, t = 1.58e12;
        # IC_blastwave::setup_cyl_bw (ics/blast_wave.cpp:555-615): cells whose centre is within the blast radius
        L = 30.86e18
        cfg = abi.make_config(2, [32, 16], abi.EQEUL, abi.FLUX_RSroe, artvisc=abi.AV_FKJ98_1D, etav=0.1,
                              gamma=1.666666666666666666666, cfl=0.3, xmin=(-L, 0.0, 0.0), xmax=(L, L, 0.0),
                              bcs=["outflow", "outflow", "reflecting", "outflow"],
                              refvec=[1.0e-22, 3.0e-10, 1.0e6, 1.0e6, 1.0e6], strict_fp=strict_fp, coord_sys=2)
        P = problems.alloc(cfg)
        X, Y, _ = problems.mesh(cfg)
        rb = 2.0 * cfg.dx
        P[abi.RO] = 2.34e-22
        P[abi.PG] = np.where(np.sqrt(X * X + Y * Y) <= rb, 3.0 * 1.0e51 * (cfg.gamma - 1.0) / (4.0 * np.pi * rb ** 3), 1.38e-11)
        return cfg, P, 1.58e12, 400
    if name == "mhdbwaxi2d_halfplane_nr032":
        # test_problems/blastwave_axi2d/params_MHDaxi2dBW_HalfPlane_NR128.txt at 64 x 32: as above with glm-mhd,
        # HLLD (7) and the shipped axial field (BWmagfieldX, converted as ics/blast_wave.cpp:118-120 does)
        L = 30.86e18
        bx = _MHDBW_BX
        cfg = abi.make_config(2, [64, 32], _MHDBW["eq"], _MHDBW["solver"], artvisc=_MHDBW["av"], etav=_MHDBW["eta"],
                              gamma=1.666666666666666666666, cfl=_MHDBW["cfl"], xmin=(-L, 0.0, 0.0), xmax=(L, L, 0.0),
                              bcs=_MHDBW["bcs"], refvec=[1.0e-22, 3.0e-10, 1.0e6, 1.0e6, 1.0e6, 1.0e-5, 1.0e-5, 1.0e-5, 1.0][
                                  :(9 if _MHDBW["eq"] == abi.EQGLM else 8)],
                              strict_fp=strict_fp, coord_sys=2)
        P = problems.alloc(cfg)
        X, Y, _ = problems.mesh(cfg)
        rb = _MHDBW["nzones"] * cfg.dx
        P[abi.RO] = _MHDBW["ro"]
        P[abi.PG] = np.where(np.sqrt(X * X + Y * Y) <= rb, 3.0 * _MHDBW["energy"] * (cfg.gamma - 1.0) / (4.0 * np.pi * rb ** 3),
                             _MHDBW["pg"])
        P[abi.BX] = bx
        return cfg, P, _MHDBW["tf"], 300
    raise KeyError(name)
