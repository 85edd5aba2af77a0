/* pion_host.h -- C view of the C++ host layer above include/pion_gpu.h (libpion_host.so, pion_amd/host/).
 *
 * The host layer mirrors the reference's CALLER side of the hot path -- the three functions of sim_control that a PION
 * build overrides and the MPI-side pieces they use:
 *   pion_host_sim_*   pion_host::sim_control_gpu   sim_control::Time_Int                sim_control/sim_control.cpp:202-281
 *                                                  calc_timestep::calculate_timestep    sim_control/calc_timestep.cpp:68-262
 *                                                  time_integrator::advance_time        sim_control/time_integrator.cpp:72-250
 *   pion_host_comm_*  pion_host::slab_comm         comm_mpi::send/receive_cell_data     comms/comm_mpi.cpp:287-425
 *                     (slab_comm_rccl: RCCL;       comm_mpi::global_operation_double    comms/comm_mpi.cpp:182-209
 *                      slab_comm_shm: host memory) MCMD_bc::BC_update_BCMPI             boundaries/MCMD_boundaries.cpp:122-237
 *   pion_host_build_cooling_tables                 mp_only_cooling::gen_mpoc_lookup_tables  microphysics/mp_only_cooling.cpp:528-579
 * C++ callers use the classes directly (sim_control_gpu.h, slab_comm.h); this view is what ctypes / a C driver binds.
 * All functions return 0 or a negative PION_GPU_* code unless stated; none exits. */
#ifndef PION_HOST_H
#define PION_HOST_H

#include "pion_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

struct pion_backend;   /* pion_amd/host/pion_backend.h: what the loop calls below itself; NULL = libpion_gpu.so */

/* ---- the time loop (one object per rank / GPU) */
int pion_host_sim_create(const pion_gpu_config *cfg, int device, void **sim);
int pion_host_sim_create_backend(const pion_gpu_config *cfg, int device, const struct pion_backend *backend, void **sim);
void pion_host_sim_destroy(void *sim);
void *pion_host_sim_handle(void *sim);                       /* the pion_gpu handle (set-up calls: wind cells, tables) */
/* sim_init::Init: upload P ([nvar][nz_all][ny_all][nx_all]), Ph = P, assign + update boundaries.  A second call on
 * the same object discards the time-step request of the previous state. */
int pion_host_sim_init(void *sim, const double *P, double simtime, double finishtime, double first_step_dt_limit);
int pion_host_sim_set_time(void *sim, int timestep, double last_dt);   /* restart: SimParams::timestep / last_dt */
int pion_host_sim_step(void *sim, double *dt);                /* calculate_timestep + advance_time */
int pion_host_sim_time_int(void *sim, int nsteps, double *simtime, double *last_dt);   /* returns steps taken, < 0 on error */
int pion_host_sim_download(void *sim, int which, double *P);
int pion_host_sim_finish_halo(void *sim);
int pion_host_sim_set_comm(void *sim, void *comm);            /* before pion_host_sim_init */
int pion_host_sim_last_error(void *sim, char *buf, int len);

/* ---- z-slab communicators (both return a pion_host::slab_comm*) */
int pion_host_comm_unique_id(void *out128);                   /* ncclGetUniqueId on rank 0 */
int pion_host_comm_create(int rank, int world, int periodic_z, const void *unique_id, int device, void **comm);   /* RCCL */
int pion_host_comm_shm_create(int rank, int world, int periodic_z, const char *name, const struct pion_backend *backend,
                              void **comm);                   /* host-staged: POSIX shared memory "/name" */
void pion_host_comm_destroy(void *comm);
int pion_host_comm_attach(void *comm, void *gpu_handle);
int pion_host_comm_start(void *comm, int which);
int pion_host_comm_finish(void *comm);
int pion_host_comm_allreduce_min(void *comm, double *t_dyn, double *t_mp);
int pion_host_comm_last_error(void *comm, char *buf, int len);

/* ---- cooling tables of mp_only_cooling (EP.cooling = 8): T[nT], tabs[5][nT] = {rrhp, C_rrh, C_ffhe, C_fbdn, C_cie},
 * slopes[5][nT]; and the three spline-backed rate curves they are built from */
int pion_host_build_cooling_tables(double min_temp, double max_temp, int nT, double *T, double *tabs, double *slopes);
double pion_host_cooling_rate_wss09(double T);
double pion_host_hii_rrr(double T);
double pion_host_hii_total_cooling(double T);

/* the product's only backend table: libpion_gpu.so */
const struct pion_backend *pion_backend_gpu(void);

#ifdef __cplusplus
}
#endif
#endif /* PION_HOST_H */
