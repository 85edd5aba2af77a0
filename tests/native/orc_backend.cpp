// orc_backend.cpp -- TEST INFRASTRUCTURE: pion_backend (pion_amd/host/pion_backend.h) bound to the CPU oracle
// (oracle/liboracle.so), so that the product's C++ time loop (pion_host::sim_control_gpu) and its host-staged slab
// transport (pion_host::slab_comm_shm) can run as two processes where no GPU is -- tests/test_host_two_ranks.py.
// Built by tests/native/Makefile into tests/native/liborc_backend.so; never loaded by the product.
#include <cstring>
#include <vector>

#include "../../pion_amd/host/pion_backend.h"

extern "C" {
int orc_create(const pion_gpu_config *cfg, void **handle);
void orc_destroy(void *h);
int orc_last_error(void *h, char *buf, int len);
int orc_ng_all(void *h, int a);
int orc_upload(void *h, const double *Psoa);
int orc_upload_which(void *h, int which, const double *Psoa);
int orc_download(void *h, int which, double *Psoa);
int orc_update_bcs(void *h, double simtime, int cstep, int maxstep, int assign);
int orc_calc_dt(void *h, double *t_dyn, double *t_mp);
int orc_set_glm_speeds(void *h, double dt, double dx, double cr);
int orc_stage(void *h, double dt, int space_ooa, int is_full_step);
}

namespace {
struct H {
  void *o;
  pion_gpu_config cfg;
  long nx, ny, nzall, ncell;
  double td, tm;
  int dt_rc;
  std::vector<double> buf;
};
int b_create(const pion_gpu_config *cfg, int, void **handle)
{
  H *h = new H;
  h->cfg = *cfg;
  *handle = h;
  const int rc = orc_create(cfg, &h->o);
  if (rc) return rc;
  h->nx = orc_ng_all(h->o, 0);
  h->ny = orc_ng_all(h->o, 1);
  h->nzall = orc_ng_all(h->o, 2);
  h->ncell = h->nx * h->ny * h->nzall;
  h->buf.resize((size_t)cfg->nvar * h->ncell);
  return 0;
}
void b_destroy(void *p)
{
  H *h = (H *)p;
  if (h->o) orc_destroy(h->o);
  delete h;
}
int b_last_error(void *p, char *buf, int len) { return orc_last_error(((H *)p)->o, buf, len); }
int b_upload(void *p, const double *P) { return orc_upload(((H *)p)->o, P); }
int b_download(void *p, int which, double *P) { return orc_download(((H *)p)->o, which, P); }
int b_update_bcs(void *p, double t, int c, int m, int a) { return orc_update_bcs(((H *)p)->o, t, c, m, a); }
int b_stage(void *p, double dt, int ooa, int full) { return orc_stage(((H *)p)->o, dt, ooa, full); }
// the oracle has no split: everything happens in the z-boundary call, after the halo has arrived
int b_stage_part(void *p, double dt, int ooa, int full, int part)
{
  return (part == PION_STAGE_INTERIOR) ? 0 : orc_stage(((H *)p)->o, dt, ooa, full);
}
int b_glm(void *p, double dt, double dx, double cr) { return orc_set_glm_speeds(((H *)p)->o, dt, dx, cr); }
int b_calc_dt(void *p, double *a, double *b) { return orc_calc_dt(((H *)p)->o, a, b); }
int b_dt_begin(void *p)
{
  H *h = (H *)p;
  h->dt_rc = orc_calc_dt(h->o, &h->td, &h->tm);
  return h->dt_rc;
}
int b_dt_wait(void *p, double *a, double *b)
{
  H *h = (H *)p;
  *a = h->td;
  *b = h->tm;
  return h->dt_rc;
}
long b_halo_count(void *p)
{
  H *h = (H *)p;
  return (h->cfg.ndim == 3) ? (long)h->cfg.nvar * h->cfg.nbc * h->nx * h->ny : 0;
}
int b_to_host_begin(void *p, int which, double *lo, double *hi)
{
  H *h = (H *)p;
  if (int rc = orc_download(h->o, which, h->buf.data())) return rc;
  const long plane = h->nx * h->ny, nb = h->cfg.nbc, nz = h->cfg.ng[2], n = nb * plane;
  for (int v = 0; v < h->cfg.nvar; v++) {
    const double *A = h->buf.data() + (size_t)v * h->ncell;
    if (lo) memcpy(lo + (size_t)v * n, A + nb * plane, n * sizeof(double));   // first on-grid planes
    if (hi) memcpy(hi + (size_t)v * n, A + nz * plane, n * sizeof(double));   // last on-grid planes
  }
  return 0;
}
int b_to_host_end(void *) { return 0; }
int b_from_host(void *p, int which, const double *lo, const double *hi)
{
  H *h = (H *)p;
  const long plane = h->nx * h->ny, nb = h->cfg.nbc, nz = h->cfg.ng[2], n = nb * plane;
  // a full step: the reference sets P = Ph in the received ghost cells (MCMD_boundaries.cpp:215-224); the
  // oracle keeps both arrays, so the planes go into both
  for (int w = which; w <= (which == 0 ? 1 : which); w++) {
    if (int rc = orc_download(h->o, w, h->buf.data())) return rc;
    for (int v = 0; v < h->cfg.nvar; v++) {
      double *A = h->buf.data() + (size_t)v * h->ncell;
      if (lo) memcpy(A, lo + (size_t)v * n, n * sizeof(double));
      if (hi) memcpy(A + (nz + nb) * plane, hi + (size_t)v * n, n * sizeof(double));
    }
    if (int rc = orc_upload_which(h->o, w, h->buf.data())) return rc;
  }
  return 0;
}
const pion_backend k_orc = {"oracle (test backend)", b_create, b_destroy, b_last_error, b_upload, b_download,
                            b_update_bcs, b_stage, b_stage_part, b_glm, b_calc_dt, b_dt_begin, b_dt_wait,
                            b_halo_count, b_to_host_begin, b_to_host_end, b_from_host};
}  // namespace

extern "C" const pion_backend *pion_backend_oracle(void) { return &k_orc; }
extern "C" void *pion_backend_oracle_handle(void *backend_handle) { return ((H *)backend_handle)->o; }
