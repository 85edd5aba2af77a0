#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from oracle/_ref/libpion_ref.so, i.e. from the
REFERENCE's own solver objects (compiled from /root/reference by `make -C oracle ref`) driven by
oracle/ref_harness.cpp.  Run in the build container only (the reference does not travel):

    make -C oracle ref && python tests/golden/make_golden.py

Fixtures are data: seeded inputs and the reference's outputs, stored as compressed .npz.
  flux_kat.npz      InterCellFlux known-answer vectors: every flux solver the path supports
                    (HD: LF, linear, exact, hybrid, Roe-CV (+H-corr eta), Roe-PV, FVS, HLL;
                     MHD and GLM-MHD: LF, HLL, HLLD incl. the HLL switch), with and without
                    tracers and FKJ98 viscosity, along all three axes, 160 state pairs each
                    including degenerate pairs (equal states, Bx=0, Bt=0, supersonic, vacuum).
  flux_kat_b.npz    the same for the Roe-MHD solver (with and without the H-correction eta) and the
                    FKJ98 linear MHD solver, ideal and GLM-MHD (own seed; `make_golden.py b`).
  steps_b.npz       whole-grid dumps for ideal-MHD Roe + H-correction 2-D, GLM-MHD Roe 3-D, GLM-MHD
                    linear solver 2-D with mixed boundaries, 3-D hydro jet (internal JETBC boundary), and five
                    2-D cylindrical (z,R) axisymmetric blasts (HD Roe, HD H-correction + tracer, ideal-MHD
                    HLLD, GLM-MHD HLLD + tracer, GLM-MHD Roe first order).
  cell_kat.npz      CellAdvanceTime and CellTimeStep vectors (incl. negative-pressure repair,
                    with and without a microphysics object).
  endstate.npz      long runs (`make_golden.py c`): low-resolution versions of BASELINE configs 1-4 (spherical 1-D
                    Sedov n128, DMR 65 x 20 to t = 0.2, MHD blast 64 x 96 to t = 0.2 as GLM-MHD and as ideal MHD,
                    3-D octant blast 32^3, 80 steps): every dt, the end state, the conserved totals.
  steps.npz         whole-grid dumps after 2 second-order steps (and per-stage aux data) for small
                    grids: HD Roe 3-D octant blast (reflecting/outflow), HD FVS 2-D with tracer,
                    HD Roe + H-correction 2-D, ideal-MHD HLLD 2-D periodic, GLM-MHD HLLD 3-D
                    periodic blast, GLM-MHD mixed outflow/one-way/reflecting 2-D, DMR 2-D, LF 1st order.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from pion_amd import abi, driver, problems  # noqa: E402
from cpu_backends import CpuSim  # noqa: E402
import golden_cases as gc  # noqa: E402


def flux_kats(rng, cases, fname):
    out = {}
    for (eq, sv, ntr, av) in cases:
        cfg = gc.flux_cfg(eq, sv, ntr, av)
        L, R = problems.random_states(rng, gc.NFLUX, eq, ntr)
        aux = np.zeros((gc.NFLUX, 4))
        if av == abi.AV_HCORR_FKJ98:
            aux[:, 0] = rng.uniform(0, 2, gc.NFLUX)
        if sv == abi.FLUX_RS_HLLD:
            aux[:, 1] = rng.uniform(0, 1, gc.NFLUX) < 0.3
        key = gc.flux_key(eq, sv, ntr, av)
        out[key + "_L"], out[key + "_R"], out[key + "_aux"] = L, R, aux
        with CpuSim(cfg, "ref") as r:
            r.set_glm_speeds(gc.GLM_DT, cfg.dx, 0.25 / cfg.dx)
            for ax in range(3):
                F, _ = r.interface_flux(ax, L, R, aux, dt=gc.GLM_DT)
                out[key + "_F%d" % ax] = F
    np.savez_compressed(os.path.join(HERE, fname), **out)


def step_dumps(cases, fname):
    out = {}
    for name in cases:
        cfg, P = gc.step_case(name)
        with CpuSim(cfg, "ref") as r:
            gc.step_setup(name, r)
            sc = driver.SimControl(r, cfg)
            sc.init(P)
            out[name + "_bc"] = r.download(0).astype(np.float64)
            dts = []
            for _ in range(gc.NSTEPS):
                dts.append(sc.calculate_timestep())
                sc.advance_time()
            out[name + "_dt"] = np.array(dts)
            out[name + "_P"] = r.download(0)
    np.savez_compressed(os.path.join(HERE, fname), **out)


def main_b():
    """flux_kat_b.npz: Roe-MHD and linear-MHD interface fluxes (added after the first set; separate
    seed and file so that the vectors of flux_kat.npz stay what they were)"""
    flux_kats(np.random.default_rng(20240612), gc.flux_cases_b(), "flux_kat_b.npz")
    step_dumps(gc.STEP_CASES_B, "steps_b.npz")
    for f in ("flux_kat_b.npz", "steps_b.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


def main():
    rng = np.random.default_rng(20240611)
    # ---- interface-flux KATs
    flux_kats(rng, gc.flux_cases(), "flux_kat.npz")

    # ---- cell KATs
    out = {}
    for (eq, ntr, cool) in gc.cell_cases():
        cfg = gc.cell_cfg(eq, ntr, cool)
        P, dU = gc.cell_inputs(rng, cfg)
        key = gc.cell_key(eq, ntr, cool)
        with CpuSim(cfg, "ref") as r:
            r.set_glm_speeds(gc.GLM_DT, cfg.dx, 0.25 / cfg.dx)
            out[key + "_P"], out[key + "_dU"] = P, dU
            out[key + "_Pf"] = r.cell_advance(P, dU, fv_dt=gc.GLM_DT)
            out[key + "_dt"] = r.cell_timestep(P)
    np.savez_compressed(os.path.join(HERE, "cell_kat.npz"), **out)

    # ---- whole-grid steps
    step_dumps(gc.STEP_CASES, "steps.npz")
    # ---- Cash-Karp integrator KAT: the REFERENCE's Integrator_Base (microphysics/integrator.cpp)
    # integrating dE/dt = f(E), f piecewise linear (see gc.ode_table)
    import ctypes as C
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libpion_ref.so"))
    xs, ys = gc.ode_table()
    E0, dts = gc.ode_inputs(rng)
    n = E0.size
    Eout, tout = np.zeros(n), np.zeros(n)
    errs = np.zeros(n, dtype=np.int32)
    dp = C.POINTER(C.c_double)
    lib.ref_integrate(C.c_int(xs.size), xs.ctypes.data_as(dp), ys.ctypes.data_as(dp), C.c_int(n),
                      E0.ctypes.data_as(dp), dts.ctypes.data_as(dp), C.c_double(1.0e-2),
                      Eout.ctypes.data_as(dp), tout.ctypes.data_as(dp), errs.ctypes.data_as(C.POINTER(C.c_int)))
    np.savez_compressed(os.path.join(HERE, "ode_kat.npz"), E0=E0, dt=dts, Eout=Eout, tout=tout, errs=errs)
    for f in ("flux_kat.npz", "cell_kat.npz", "steps.npz", "ode_kat.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


def main_c():
    """endstate.npz: low-resolution versions of BASELINE configs 1-4 run by the reference objects to their
    finish time (or a fixed number of steps): every dt taken, the end state (on-grid and ghosts), the
    conserved totals.  `make_golden.py c`."""
    import time
    out = {}
    for name in gc.END_CASES:
        cfg, P, tf, nmax = gc.end_case(name)
        t0 = time.time()
        with CpuSim(cfg, "ref") as r:
            n, t, dts = gc.end_run(r, cfg, P, tf, nmax)
            A = r.download(0)
        tot, _ = gc.conserved_totals(cfg, A)
        out[name + "_n"], out[name + "_t"], out[name + "_dt"] = np.array(n), np.array(t), dts
        out[name + "_P"], out[name + "_tot"] = A, tot
        print("%-22s %4d steps to t = %.6g  (%.1f s)" % (name, n, t, time.time() - t0))
    np.savez_compressed(os.path.join(HERE, "endstate.npz"), **out)
    print("endstate.npz", os.path.getsize(os.path.join(HERE, "endstate.npz")) // 1024, "KiB")


def main_d():
    """cooling_kat.npz, steps_c.npz, endstate_c.npz: the REFERENCE's own mp_only_cooling (compiled from
    microphysics/mp_only_cooling.cpp; oracle/ref_cooling.cpp) -- its look-up tables, Edot, TimeUpdateMP and
    timescales on seeded inputs, and whole steps / a 60-step run of the cooling configuration with it as the
    global MP object.  The three spline-backed rate curves are supplied by this script from the product's table
    builder (their values are the part that stays parity-unpinned).  `make_golden.py d`."""
    from cpu_backends import RefCooling, install_ref_rate_curves
    from pion_amd import cooling
    rng = np.random.default_rng(20240613)
    out = {}
    for k, (tmin, tmax) in enumerate(gc.COOL_RANGES):
        cfg = gc.cool_cfg(k)
        with RefCooling(tmin, tmax, gc.COOL_GAMMA, gc.COOL_NVAR, 1) as r, CpuSim(cfg, "orc") as o:
            T, tabs, sl = r.tables()
            o.set_cooling_tables(T, tabs, sl)
            key = "r%d_" % k
            out[key + "T"], out[key + "tabs"], out[key + "slopes"], out[key + "limits"] = T, tabs, sl, r.limits()
            rho, Te = gc.cool_edot_inputs(rng, tmin, tmax, T)
            out[key + "edot_rho"], out[key + "edot_T"], out[key + "edot"] = rho, Te, r.edot(rho, Te)
            P, _ = gc.cool_states(rng, tmin, tmax)
            out[key + "P"] = P
            out[key + "tcool"] = r.timescale(P)
            for j, dt in enumerate(gc.COOL_DTS):
                # the reference exits the process when its integrator gives up: keep the states the oracle
                # (screened one by one) integrates, and let the reference run exactly those
                ok = np.zeros(P.shape[0], dtype=bool)
                for i in range(P.shape[0]):
                    try:
                        o.cooling_update(P[i:i + 1], dt)
                        ok[i] = True
                    except RuntimeError:
                        pass
                Pout, Tf = r.update(P[ok], dt)
                out[key + "upd%d_ok" % j], out[key + "upd%d_P" % j], out[key + "upd%d_Tf" % j] = ok, Pout, Tf
                print("range %d dt %.0e: %d of %d states integrated" % (k, dt, ok.sum(), ok.size))
    np.savez_compressed(os.path.join(HERE, "cooling_kat.npz"), **out)
    # whole steps and a long run with the reference's mp_only_cooling as MP
    install_ref_rate_curves()
    out = {}
    for name in gc.STEP_CASES_C:
        cfg, P = gc.step_case_c(name)
        with CpuSim(cfg, "ref") as r:
            sc = driver.SimControl(r, cfg)
            sc.init(P)
            dts = []
            for _ in range(gc.NSTEPS):
                dts.append(sc.calculate_timestep())
                sc.advance_time()
            out[name + "_dt"] = np.array(dts)
            out[name + "_P"] = r.download(0)
    np.savez_compressed(os.path.join(HERE, "steps_c.npz"), **out)
    out = {}
    for name in gc.END_CASES_C:
        cfg, P, tf, nmax = gc.end_case_c(name)
        with CpuSim(cfg, "ref") as r:
            n, t, dts = gc.end_run(r, cfg, P, tf, nmax)
            A = r.download(0)
        tot, _ = gc.conserved_totals(cfg, A)
        out[name + "_n"], out[name + "_t"], out[name + "_dt"] = np.array(n), np.array(t), dts
        out[name + "_P"], out[name + "_tot"] = A, tot
        print("%-22s %4d steps to t = %.6g" % (name, n, t))
    np.savez_compressed(os.path.join(HERE, "endstate_c.npz"), **out)
    for f in ("cooling_kat.npz", "steps_c.npz", "endstate_c.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


def main_e():
    """shocktubes.npz: the shock-tube initial conditions the reference ships (gc.SHOCK_TUBES: Toro 1-5, Brio-Wu,
    Falle FS/SS/FR/SR/OFS, Ryu-Jones 1a-5b) run by the reference objects in 1-D to their finish times: every dt and
    the end state.  One child process per case: the reference exits the process on a failed Riemann solve.
    `make_golden.py e [case]`."""
    import subprocess
    if len(sys.argv) > 2:
        name = sys.argv[2]
        cfg, P, tf = gc.shock_tube_case(name)
        with CpuSim(cfg, "ref") as r:
            n, t, dts = gc.end_run(r, cfg, P, tf, 100000)
            A = r.download(0)
        np.savez(os.path.join("/tmp", "st_%s.npz" % name), n=n, t=t, dt=dts, P=A)
        return
    out = {}
    for name in gc.SHOCK_TUBES:
        rc = subprocess.call([sys.executable, os.path.abspath(__file__), "e", name], stdout=subprocess.DEVNULL)
        if rc != 0:
            print("%-10s the reference gave up (exit %d): no fixture" % (name, rc))
            continue
        z = np.load(os.path.join("/tmp", "st_%s.npz" % name))
        for k in ("n", "t", "dt", "P"):
            out[name + "_" + k] = z[k]
        print("%-10s %4d steps to t = %.6g" % (name, int(z["n"]), float(z["t"])))
    np.savez_compressed(os.path.join(HERE, "shocktubes.npz"), **out)
    print("shocktubes.npz", os.path.getsize(os.path.join(HERE, "shocktubes.npz")) // 1024, "KiB")


def main_f():
    """endstate_s.npz: more of the reference's shipped uniform-grid test problems (gc.END_CASES_S: FieldLoop x 3,
    advection of a contact discontinuity, Liska-Wendroff implosion, oblique shocks M25 / M40 with three solver /
    viscosity pairs, the axisymmetric blast waves in Euler and glm-mhd) run by the reference objects: every dt, the end
    state, the conserved totals.  One child process per case (a failed Riemann solve ends the process).
    `make_golden.py f [case]`."""
    import subprocess
    import time
    if len(sys.argv) > 2:
        name = sys.argv[2]
        cfg, P, tf, nmax = gc.end_case_s(name)
        with CpuSim(cfg, "ref") as r:
            n, t, dts = gc.end_run(r, cfg, P, tf, nmax)
            A = r.download(0)
        tot, _ = gc.conserved_totals(cfg, A)
        np.savez(os.path.join("/tmp", "es_%s.npz" % name), n=n, t=t, dt=dts, P=A, tot=tot)
        return
    out = {}
    for name in gc.END_CASES_S:
        t0 = time.time()
        rc = subprocess.call([sys.executable, os.path.abspath(__file__), "f", name], stdout=subprocess.DEVNULL)
        if rc != 0:
            print("%-28s the reference gave up (exit %d): no fixture" % (name, rc))
            continue
        z = np.load(os.path.join("/tmp", "es_%s.npz" % name))
        for k in ("n", "t", "dt", "P", "tot"):
            out[name + "_" + k] = z[k]
        print("%-28s %4d steps to t = %.6g  (%.1f s)" % (name, int(z["n"]), float(z["t"]), time.time() - t0))
    np.savez_compressed(os.path.join(HERE, "endstate_s.npz"), **out)
    print("endstate_s.npz", os.path.getsize(os.path.join(HERE, "endstate_s.npz")) // 1024, "KiB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "f":
        main_f()
    elif len(sys.argv) > 1 and sys.argv[1] == "e":
        main_e()
    elif len(sys.argv) > 1 and sys.argv[1] == "d":
        main_d()
    elif len(sys.argv) > 1 and sys.argv[1] == "b":
        main_b()
    elif len(sys.argv) > 1 and sys.argv[1] == "c":
        main_c()
    else:
        main()
        main_b()
        main_c()
        main_d()
        main_e()
        main_f()
