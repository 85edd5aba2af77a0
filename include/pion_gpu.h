/*
 * pion_gpu.h -- C-ABI of libpion_gpu.so: the MI355X (gfx950) replacement for
 * PION's finite-volume flux-update hot path.
 *
 * Every entry point is stage-granular: it replaces one of the loops that the
 * reference's time_integrator / calc_timestep run cell-by-cell through the
 * virtual FV_solver_base interface.  The reference interface each function
 * replaces is cited as file:line relative to the PION source tree (source/).
 *
 * Conventions (mirroring the reference):
 *   - pion_flt is double (defines/functionality_flags.h: PION_DATATYPE_DOUBLE).
 *   - primitive vector P = {RO,PG,VX,VY,VZ,BX,BY,BZ,SI,tracers...},
 *     conserved U = {RHO,ERG,MMX,MMY,MMZ,BBX,BBY,BBZ,PSI,...}  (constants.h:256-281).
 *   - all functions return 0 on success or a negative PION_GPU_E* code; they
 *     never exit().  Fatal physics conditions of the reference (rep.error ->
 *     exit(1), tools/reporting.h:57-70) come back as PION_GPU_EPHYSICS and a
 *     text from pion_gpu_last_error().
 *   - host arrays are owned by the caller, device memory by the handle.
 *   - a handle is not thread-safe (like the reference's solver object,
 *     solver_eqn_base.h:52), distinct handles may be used from distinct threads.
 *
 * Grid layout handed over the boundary ("SoA"): double [nvar][nz_all][ny_all][nx_all],
 * x fastest, including nbc ghost cells on every used axis, i.e. the order in
 * which UniformGrid numbers its cells (grid/uniform_grid.cpp:482-636, id =
 * ix + nx_all*(iy + ny_all*iz) counted from the most negative ghost corner),
 * with the state-vector index as the slowest index.  Unused axes have extent 1
 * and no ghosts.
 */
#ifndef PION_GPU_H
#define PION_GPU_H

#ifdef __cplusplus
extern "C" {
#endif

#define PION_MAX_NVAR 16
#define PION_MAX_DIM 3

/* equation types (constants.h:163-170) */
#define PION_EQEUL 1
#define PION_EQMHD 2
#define PION_EQGLM 3

/* flux solvers (constants.h:238-246) */
#define PION_FLUX_LF 0
#define PION_FLUX_RSlinear 1
#define PION_FLUX_RSexact 2
#define PION_FLUX_RShybrid 3
#define PION_FLUX_RSroe 4
#define PION_FLUX_RSroe_pv 5
#define PION_FLUX_FVS 6
#define PION_FLUX_RS_HLLD 7
#define PION_FLUX_RS_HLL 8

/* artificial viscosity (constants.h:321-326) */
#define PION_AV_NONE 0
#define PION_AV_FKJ98_1D 1
#define PION_AV_HCORRECTION 3
#define PION_AV_HCORR_FKJ98 4

/* boundary types handled on the device (boundaries/boundaries.h) */
#define PION_BC_PERIODIC 1
#define PION_BC_OUTFLOW 2
#define PION_BC_INFLOW 3
#define PION_BC_REFLECTING 4
#define PION_BC_FIXED 5
#define PION_BC_ONEWAY_OUT 6
#define PION_BC_DMACH 7   /* YP boundary of the double Mach reflection test */
#define PION_BC_DMACH2 8  /* internal: fixed post-shock state in y<0, x<=1/6 */
#define PION_BC_STWIND 9  /* internal: stellar-wind cells (fixed per-cell state) */
#define PION_BC_SLAB 10   /* z face owned by a neighbouring GPU (halo exchange) */
#define PION_BC_JET 11    /* internal: jet inflow cells on the XN face (pion_gpu_set_jet) */
#define PION_BC_JETREFLECT 13   /* reflecting wall behind a jet: v_n and the TANGENTIAL field change sign
                                 * (jetreflect_boundaries.cpp:32-62) */
#define PION_BC_AXISYMMETRIC 12 /* R = 0 axis of a cylindrical (z,R) grid, face YN only (axisymmetric_boundaries.cpp) */

/* cooling functions of mp_only_cooling (microphysics/mp_only_cooling.h) */
#define PION_COOL_NONE 0
#define PION_COOL_WSS09_CIE_LINE_HEAT_COOL 8

/* cell flag bits (grid/cell_interface.h:83-121) */
#define PION_CELL_ISGD 1
#define PION_CELL_ISBD 2
#define PION_CELL_ISDOMAIN 4
#define PION_CELL_TIMESTEP 8
#define PION_CELL_ISLEAF 16

/* error codes */
#define PION_GPU_OK 0
#define PION_GPU_EINVAL (-1)   /* bad argument / unsupported configuration */
#define PION_GPU_EDEVICE (-2)  /* HIP runtime error */
#define PION_GPU_EPHYSICS (-3) /* negative density etc. (reference: rep.error) */
#define PION_GPU_ENOMEM (-4)

/*
 * The subset of SimParams (sim_params.h:200-285) the hot path reads.
 */
typedef struct pion_gpu_config {
  int ndim;      /* SimParams::ndim */
  int nvar;      /* SimParams::nvar (includes tracers) */
  int ntracer;   /* SimParams::ntracer; tracers are the last ntracer variables */
  int eqntype;   /* PION_EQ* */
  int solver;    /* SimParams::solverType, PION_FLUX_* */
  int artvisc;   /* SimParams::artviscosity, PION_AV_* */
  int sp_ooa;    /* SimParams::spOOA */
  int tm_ooa;    /* SimParams::tmOOA */
  int coord_sys; /* 1 = Cartesian; 2 = cylindrical (z,R), 2-D axisymmetric: x axis = z, y axis = R
                  * (coord_sys/VectorOps.cpp:662-1245 and the cyl_FV_solver_* classes); 3 = spherical
                  * symmetry, 1-D, Euler only (VectorOps_spherical.cpp, sph_FV_solver_Hydro_Euler) */
  int nbc;       /* ghost depth, SimParams::Nbc (2 for second order) */
  int ng[PION_MAX_DIM];      /* on-grid cells per axis (1 on unused axes) */
  double xmin[PION_MAX_DIM]; /* physical position of the low corner of the ON-GRID region */
  double dx;                 /* cell size */
  double gamma;              /* SimParams::gamma */
  double cfl;                /* SimParams::CFL */
  double etav;               /* SimParams::etav */
  double min_temp;           /* EP.MinTemperature */
  double max_temp;           /* EP.MaxTemperature */
  double refvec[PION_MAX_NVAR]; /* SimParams::RefVec */
  int bc_type[6];            /* PION_BC_* for XN,XP,YN,YP,ZN,ZP (0 on unused axes) */
  int bc_dmach2;             /* 1: internal DMR2 boundary active */
  int cooling;               /* EP.cooling (PION_COOL_*), 0 = no microphysics object */
  int mp_timestep_limit;     /* EP.MP_timestep_limit: 0 none; 1,2,3 cooling time; 4 none (recombination only); else EINVAL */
  int strict_fp;             /* 1: kernels built without FMA contraction (bit-parity build) */
} pion_gpu_config;

/* ---- lifetime --------------------------------------------------------- */

/* setup_fixed_grid::set_equations (grid/setup_fixed_grid.cpp:1067-1191) +
 * setup_grid (:161-245): creates solver state and device arrays on `device`. */
int pion_gpu_create(const pion_gpu_config *cfg, int device, void **handle);
void pion_gpu_destroy(void *handle);
int pion_gpu_last_error(void *handle, char *buf, int len);

/* total cells including ghosts, and extents with ghosts */
long pion_gpu_ncell_all(void *handle);
int pion_gpu_ng_all(void *handle, int axis);

/* ---- state transfer ---------------------------------------------------- */

/* dataio->ReadData + "Ph=P" (sim_control/sim_init.cpp:219-241): copies a host
 * SoA array into P and Ph.  Ghost values in the input are ignored once
 * pion_gpu_update_bcs has run. */
int pion_gpu_upload(void *handle, const double *P_soa);
/* which = 0: P, 1: Ph */
int pion_gpu_download(void *handle, int which, double *P_soa);
/* adopt caller-owned device buffers (e.g. torch tensors) instead of internal ones;
 * both must hold nvar*ncell_all doubles */
int pion_gpu_bind_device_state(void *handle, void *dP, void *dPh);
void *pion_gpu_device_ptr(void *handle, int which);
/* HIP stream all subsequent work of this handle is issued on (hipStream_t) */
int pion_gpu_set_stream(void *handle, void *stream);
/* Second HIP stream for pion_gpu_pack_halo / pion_gpu_unpack_halo (NULL: the compute stream).
 * The library orders pack after the compute stream's work so far and PION_STAGE_ZBOUNDARY after the
 * last unpack; the caller's transfer (RCCL/MPI) must be enqueued on, or ordered with, this stream. */
int pion_gpu_set_comm_stream(void *handle, void *stream);
/* the handle's streams (hipStream_t): which = 0 compute, 1 comm */
void *pion_gpu_get_stream(void *handle, int which);
int pion_gpu_synchronize(void *handle);

/* internal fixed-state cells: stellar wind (grid/stellar_wind_BC.cpp:642-677,
 * boundaries/stellar_wind_boundaries.cpp:244-350).  idx = cell ids (with
 * ghosts), states = n*nvar doubles (cell-major).  Marks the cells
 * isbd=true,isdomain=false (stellar_wind_BC.cpp:277-278). */
int pion_gpu_set_wind_cells(void *handle, long n, const long *idx, const double *states);

/* jet_bc::BC_assign_JETBC / BC_update_JETBC (boundaries/jet_boundaries.cpp:36-208, 3-D Cartesian
 * branch :170-201, update :212-262) with JetParams (sim_params.h:331-341): every XN ghost cell of an
 * on-grid (y,z) column whose centre lies within jetradius*dx of the x axis holds `jetstate`
 * (rho, p_g, v, then tracers), re-imposed after the external boundaries at every boundary update.
 * 3-D Cartesian (Euler only, as in the reference) or 2-D cylindrical: there the first jetradius rows above
 * the axis, with B = (jetstate[BX], 0, jetstate[BY]) for MHD (:74-83; the radial JETPROFILE of the
 * assignment is overwritten by the uniform state in every update, so it is not reproduced). */
int pion_gpu_set_jet(void *handle, int jetradius, const double *jetstate);

/* mp_only_cooling look-up tables (microphysics/mp_only_cooling.cpp:528-579):
 * nT temperatures, 5 value tables and 5 slope tables in the order
 * rrhp, C_rrh, C_ffhe, C_fbdn, C_cie.  2 <= nT <= 256 (the reference builds 200 points; the cooling kernel
 * keeps the tables in LDS), else PION_GPU_EINVAL. */
int pion_gpu_set_cooling_tables(void *handle, int nT, const double *T,
                                const double *tabs, const double *slopes);

/* ---- the hot path ------------------------------------------------------ */

/* assign_update_bcs::TimeUpdateInternalBCs + TimeUpdateExternalBCs
 * (boundaries/assign_update_bcs.cpp:134-252): fills ghost cells of Ph, and of
 * P too when cstep==maxstep.  `assign`!=0 additionally captures the constant
 * inflow/fixed reference states from P (BC_assign_*, inflow_boundaries.cpp,
 * fixed_boundaries.cpp) and must be used for the first call after upload. */
int pion_gpu_update_bcs(void *handle, double simtime, int cstep, int maxstep, int assign);

/* calc_timestep::calc_dynamics_dt (sim_control/calc_timestep.cpp:271-333) and
 * get_mp_timescales_no_radiation (:405-507): min over on-grid cells of
 * FV_solver_*::CellTimeStep (solver_eqn_hydro_adi.cpp:460-502,
 * solver_eqn_mhd_adi.cpp:516-582) and MP->timescales
 * (mp_only_cooling.cpp:333-368).  No limiting is applied here. */
int pion_gpu_calc_dt(void *handle, double *t_dyn, double *t_mp);
/* The same reduction in two halves, for slab-decomposed runs (sim_control_MPI.cpp:503-504 does
 * COMM->global_operation_double("MIN", .) on the host value; here the minimum stays on the device
 * until it has been reduced over the ranks):
 *   pion_gpu_calc_dt_device  enqueues the reduction (nothing, when the last full stage left the minima
 *                            behind) and returns the device address of {min t_dyn, min t_mp} (2 doubles);
 *                            no host synchronisation.  The caller may all-reduce(min) that buffer in place
 *                            on the handle's compute stream (ncclAllReduce(ncclMin)).
 *   pion_gpu_read_dt         the single 16-byte read-back + the device error word. */
int pion_gpu_calc_dt_device(void *handle, void **dptr);
int pion_gpu_read_dt(void *handle, double *t_dyn, double *t_mp);
/* pion_gpu_read_dt in two halves: _request enqueues the copy of the minima and of the error word into pinned
 * host memory (compute stream) and returns; _wait blocks until it has arrived.  A host loop requests the
 * minima right after the full-step stage, enqueues the boundary update and the halo exchange, and only then
 * waits: the read-back latency hides under work the next step needs anyway. */
int pion_gpu_dt_request(void *handle);
int pion_gpu_dt_wait(void *handle, double *t_dyn, double *t_mp);

/* FV_solver_mhd_mixedGLM_adi::Set_GLM_Speeds (solver_eqn_mhd_adi.cpp:906-922):
 * c_h = CFL*dx/dt, c_r = cr. */
int pion_gpu_set_glm_speeds(void *handle, double dt, double dx, double cr);

/* One stage of time_integrator::first_order_update / second_order_update
 * (sim_control/time_integrator.cpp:151-250) without the boundary update:
 *   Setdt(dt_stage); calc_microphysics_dU (:253-296,438-489);
 *   calc_dynamics_dU = preprocess_data + set_dynamics_dU (:498-873);
 *   grid_update_state_vector (:881-958).
 * space_ooa: OA1 (first half step) or OA2; is_full_step: step==ooa (P=Ph). */
int pion_gpu_stage(void *handle, double dt_stage, int space_ooa, int is_full_step);

/* The same stage in two parts, so that a z-slab's halo exchange (MCMD_boundaries.cpp:122-237, which
 * the reference completes before calc_dynamics_dU starts) runs underneath most of the work:
 *   PION_STAGE_INTERIOR   the on-grid planes that read no z ghost plane; may be issued while the
 *                         z halo of the stencil array is still in flight;
 *   PION_STAGE_ZBOUNDARY  the nbc planes next to each z face; ordered after the last
 *                         pion_gpu_unpack_halo (comm stream) inside the library.
 * INTERIOR followed by ZBOUNDARY gives bit for bit the result of PION_STAGE_WHOLE (= pion_gpu_stage).
 * Configurations the split does not cover (1-D/2-D, H-correction, first-order scheme, <= 2*nbc
 * planes) do all the work in the ZBOUNDARY call. */
#define PION_STAGE_WHOLE 0
#define PION_STAGE_INTERIOR 1
#define PION_STAGE_ZBOUNDARY 2
int pion_gpu_stage_part(void *handle, double dt_stage, int space_ooa, int is_full_step, int part);

/* time_integrator::advance_time (time_integrator.cpp:72-142) for OA1/OA1 and
 * OA2/OA2: stages + boundary updates. */
int pion_gpu_advance_time(void *handle, double dt, double simtime);

/* ---- slab decomposition (replaces decomposition/MCMD_control.cpp:231-309 and
 * comms/comm_mpi.cpp:287-636 for this path) -------------------------------- */

/* Copy the nbc on-grid z-planes adjacent to face (4=ZN, 5=ZP) of `which`
 * (0=P,1=Ph) into a contiguous device buffer [nvar][nbc][ny_all][nx_all], or
 * write such a buffer into the ghost planes of that face. */
long pion_gpu_halo_count(void *handle); /* doubles per halo buffer */
/* In-place exchange (no pack / unpack kernels): in the [nvar][nz_all][ny_all][nx_all] layout the nbc planes
 * next to a z face are ONE contiguous run of count_per_var doubles per variable, so a transport can send
 * from, and receive into, the state array directly (variable v: pointer + v * var_stride doubles):
 *   send_lo / send_hi  the first / last nbc on-grid planes (x/y ghosts included)
 *   recv_lo / recv_hi  the ZN / ZP ghost planes.
 * pion_gpu_halo_begin orders the communication stream after the compute stream's work so far (what
 * pion_gpu_pack_halo does first), pion_gpu_halo_end marks the point of the communication stream after
 * which the ghost planes are complete (what pion_gpu_unpack_halo does last): PION_STAGE_ZBOUNDARY and
 * whole stages wait for it inside the library. */
typedef struct {
  double *send_lo, *send_hi, *recv_lo, *recv_hi;
  long count_per_var, var_stride;
  int nvar;
} pion_gpu_halo_spans_t;
int pion_gpu_halo_spans(void *handle, int which, pion_gpu_halo_spans_t *out);
int pion_gpu_halo_begin(void *handle);
int pion_gpu_halo_end(void *handle);
int pion_gpu_pack_halo(void *handle, int which, int face, void *dbuf);
int pion_gpu_unpack_halo(void *handle, int which, int face, void *dbuf);

/* ---- test seams -------------------------------------------------------- */

/* FV_solver_base::InterCellFlux (spatial_solvers/solver_eqn_base.cpp:152-204)
 * for n independent interfaces along `axis`.  Pl, Pr: n*nvar (interface-major)
 * edge states; aux: n*4 doubles {HC_etamax, use_HLL(0/1), unused, unused};
 * F, Pstar: n*nvar outputs.  dt is the value of FV_dt (Lax-Friedrichs only). */
int pion_gpu_interface_flux(void *handle, int n, int axis, double dt, const double *Pl,
                            const double *Pr, const double *aux, double *F, double *Pstar);

/* mp_only_cooling::TimeUpdateMP (microphysics/mp_only_cooling.cpp:167-218) for n
 * independent cells: P_in n*nvar, P_out n*nvar. */
int pion_gpu_cooling_update(void *handle, int n, double dt, const double *P_in, double *P_out);
/* mp_only_cooling::Edot (:491-521) for n (rho,T) pairs */
int pion_gpu_cooling_edot(void *handle, int n, const double *rho, const double *T, double *edot);
/* mp_only_cooling::timescales(P, gamma, tc=true, ...) (:333-368) for n independent cells: P_in n*nvar,
 * t_cool n doubles (1e99 below 1.1 MinT_allowed) */
int pion_gpu_cooling_timescale(void *handle, int n, const double *P_in, double *t_cool);

/* last kernel timings, milliseconds, measured with HIP events on the handle's stream:
 * out[0]=stage kernel, out[1]=prepass, out[2]=bc fill, out[3]=dt reduction (mean per launch);
 * with n >= 8 also out[4..7] = the number of launches each mean was taken over */
int pion_gpu_enable_timing(void *handle, int on);
int pion_gpu_get_timing(void *handle, double *out, int n);

#ifdef __cplusplus
}
#endif
#endif /* PION_GPU_H */
