// pion_backend.h -- the calls the host time loop (sim_control_gpu) and the staged slab transport
// (slab_comm_shm) make below themselves, as a table of function pointers.
//
// The product has ONE implementation: pion_backend_gpu(), bound to the C-ABI of libpion_gpu.so
// (include/pion_gpu.h) -- there is no CPU implementation in the product.  The table exists so that the C++ time
// loop and its transport can be rehearsed where no GPU is (tests/native/orc_backend.cpp binds it to the test
// oracle for the world_size-2 CPU test), the way the reference's sim_control is written against the virtual
// FV_solver_base / GridBaseClass interfaces rather than one concrete class.
#ifndef PION_BACKEND_H
#define PION_BACKEND_H

#include "../../include/pion_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pion_backend {
  const char *name;
  int (*create)(const pion_gpu_config *cfg, int device, void **handle);
  void (*destroy)(void *handle);
  int (*last_error)(void *handle, char *buf, int len);
  int (*upload)(void *handle, const double *P_soa);
  int (*download)(void *handle, int which, double *P_soa);
  int (*update_bcs)(void *handle, double simtime, int cstep, int maxstep, int assign);
  int (*stage)(void *handle, double dt_stage, int space_ooa, int is_full_step);
  int (*stage_part)(void *handle, double dt_stage, int space_ooa, int is_full_step, int part);
  int (*set_glm_speeds)(void *handle, double dt, double dx, double cr);
  /* time step: calc_dt = reduce and read back now; dt_begin / dt_wait = the same, split so that the read-back of
   * the minima a full-step stage left behind overlaps the boundary update that follows it */
  int (*calc_dt)(void *handle, double *t_dyn, double *t_mp);
  int (*dt_begin)(void *handle);
  int (*dt_wait)(void *handle, double *t_dyn, double *t_mp);
  /* z-slab halos through host memory (staged transports).  One face = nvar runs of halo_count()/nvar doubles
   * (the nbc planes next to the face, full x-y extent, variable slowest).
   *   halo_to_host_begin: start copying the ON-GRID planes next to the ZN face (lo) and / or the ZP face (hi) of
   *                       array `which` (0 = P, 1 = Ph) into host buffers (either may be null); returns at once
   *   halo_to_host_end:   the host buffers are complete
   *   halo_from_host:     copy host buffers into the GHOST planes of the ZN (lo) / ZP (hi) face; the z-boundary
   *                       part of the next stage is ordered after it (the host buffers may be reused on return) */
  long (*halo_count)(void *handle);
  int (*halo_to_host_begin)(void *handle, int which, double *lo, double *hi);
  int (*halo_to_host_end)(void *handle);
  int (*halo_from_host)(void *handle, int which, const double *lo, const double *hi);
} pion_backend;

/* the product's backend: libpion_gpu.so */
const pion_backend *pion_backend_gpu(void);

#ifdef __cplusplus
}
#endif
#endif
