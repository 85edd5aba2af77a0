// sim_control_gpu.cpp -- see sim_control_gpu.h
#include "sim_control_gpu.h"

#include "slab_comm.h"

#include <algorithm>
#include <cstdio>
#include <stdexcept>
#include <string>

namespace pion_host {

sim_control_gpu::sim_control_gpu(const pion_gpu_config &c, int device, const pion_backend *backend)
    : cfg(c), be_(backend ? backend : pion_backend_gpu()), h_(nullptr)
{
  const int rc = be_->create(&cfg, device, &h_);
  if (rc != 0) {
    std::string m = h_ ? last_error() : std::string("invalid configuration");
    if (h_) be_->destroy(h_);
    h_ = nullptr;
    throw std::runtime_error(std::string(be_->name) + ": create failed (" + std::to_string(rc) + "): " + m);
  }
}
sim_control_gpu::~sim_control_gpu()
{
  if (h_) be_->destroy(h_);
}
std::string sim_control_gpu::last_error() const
{
  char buf[512] = {0};
  be_->last_error(h_, buf, sizeof buf);
  return buf;
}

int sim_control_gpu::set_comm(slab_comm *c)
{
  comm_ = c;
  return c ? c->attach(h_) : 0;
}

// TimeUpdateInternalBCs + TimeUpdateExternalBCs, then (slab runs) the start of BC_update_BCMPI for the
// array that was just written; stage() completes it between its two parts
int sim_control_gpu::update_boundaries(int cstep, int maxstep, int assign)
{
  int err = be_->update_bcs(h_, T.simtime, cstep, maxstep, assign);
  if (comm_ && !err) err += comm_->start(cstep == maxstep ? 0 : 1);
  return err;
}

int sim_control_gpu::stage(double dt, int space_ooa, int is_full)
{
  if (!comm_) return be_->stage(h_, dt, space_ooa, is_full);
  int err = be_->stage_part(h_, dt, space_ooa, is_full, PION_STAGE_INTERIOR);
  err += comm_->finish();
  err += be_->stage_part(h_, dt, space_ooa, is_full, PION_STAGE_ZBOUNDARY);
  return err;
}

int sim_control_gpu::finish_halo() { return comm_ ? comm_->finish() : 0; }

// The next step's time step depends on the state the full-step stage has just written (on-grid cells only,
// so not on the boundary update that follows): start its reduction / read-back now, wait for it in
// calculate_timestep -- the boundary kernels and the halo exchange are queued behind it meanwhile.
int sim_control_gpu::request_next_dt()
{
  if (comm_) return comm_->request_min();
  const int err = be_->dt_begin(h_);
  dt_requested_ = (err == 0);
  return err;
}

int sim_control_gpu::Init(const double *P_soa, double simtime)
{
  T.simtime = simtime;
  // a second Init on the same object (restart, new problem): the time-step read-back requested at the end of
  // the last advance_time() belongs to the old state -- discard it here, in the communicator and (upload) in
  // the library, so that calculate_timestep() reduces the state uploaded now
  dt_requested_ = false;
  if (comm_) comm_->reset();
  int err = be_->upload(h_, P_soa);
  // assign_boundary_data + TimeUpdateInternalBCs/ExternalBCs (sim_init.cpp:246-267)
  err += update_boundaries(cfg.tm_ooa, cfg.tm_ooa, 1);
  return err;
}

int sim_control_gpu::calculate_timestep()
{
  double t_dyn = 0.0, t_mp = 0.0;
  // slab runs: the minima stay on the device until they are reduced over the ranks
  // (COMM->global_operation_double("MIN", .), sim_control_MPI.cpp:503-504)
  int err;
  if (comm_) err = comm_->allreduce_min(&t_dyn, &t_mp);
  else if (dt_requested_) err = be_->dt_wait(h_, &t_dyn, &t_mp);
  else err = be_->calc_dt(h_, &t_dyn, &t_mp);
  dt_requested_ = false;
  if (err) return err;
  if (T.timestep == 0 && T.first_step_dt_limit > 0.0) t_dyn = std::min(t_dyn, T.first_step_dt_limit);
  T.dt = std::min(t_dyn, t_mp);
  // Set_GLM_Speeds(td, dx, 0.25/dx) with the *dynamical* step (calc_timestep.cpp:119-131)
  if (cfg.eqntype == PION_EQGLM) err += be_->set_glm_speeds(h_, t_dyn, cfg.dx, 0.25 / cfg.dx);
  // timestep_checking_and_limiting (calc_timestep.cpp:219-262)
  if (T.dt < T.min_timestep) throw std::runtime_error("Timestep too short!");
  T.dt = std::min(T.dt, 1.3 * T.last_dt);  // TIMESTEP_LIMITING
  T.dt = std::min(T.dt, T.finishtime - T.simtime);
  if (T.dt <= 0.0) throw std::runtime_error("Negative timestep!");
  return err;
}

int sim_control_gpu::first_order_update(double dt, int ooa)
{
  // Setdt, calc_microphysics_dU, calc_dynamics_dU(OA1), grid_update_state_vector(dt, OA1, ooa)
  return stage(dt, 1, ooa == 1 ? 1 : 0);
}
int sim_control_gpu::second_order_update(double dt, int)
{
  return stage(dt, 2, 1);
}

double sim_control_gpu::advance_time()
{
  int err = 0;
  if (cfg.tm_ooa == 1 && cfg.sp_ooa == 1) {
    err += first_order_update(T.dt, cfg.tm_ooa);
    err += request_next_dt();
    err += update_boundaries(1, 1, 0);
  }
  else if (cfg.tm_ooa == 2 && cfg.sp_ooa == 2) {
    err += first_order_update(0.5 * T.dt, 2);
    err += update_boundaries(1, 2, 0);
    err += second_order_update(T.dt, 2);
    err += request_next_dt();
    err += update_boundaries(2, 2, 0);
  }
  else throw std::runtime_error("Bad OOA requests; choose (1,1) or (2,2)");
  if (err) throw std::runtime_error("advance_time: " + last_error());
  T.simtime += T.dt;
  T.last_dt = T.dt;
  T.timestep++;
  return T.dt;
}

int sim_control_gpu::Time_Int(int nsteps)
{
  int n = 0;
  while (T.simtime < T.finishtime && (nsteps < 0 || n < nsteps)) {
    int err = calculate_timestep();
    if (err) throw std::runtime_error("calculate_timestep: " + last_error());
    advance_time();
    n++;
  }
  if (finish_halo()) throw std::runtime_error("finish_halo: " + last_error());
  return n;
}

}  // namespace pion_host

// ---- C view of the adapter (used by tests/bench through ctypes) -------------------------------
static thread_local std::string g_last_exception;
extern "C" {
// backend: null = libpion_gpu.so (the product); tests hand in another table (pion_backend.h)
int pion_host_sim_create_backend(const pion_gpu_config *cfg, int device, const pion_backend *backend, void **sim)
{
  try {
    *sim = new pion_host::sim_control_gpu(*cfg, device, backend);
    return 0;
  }
  catch (const std::exception &e) {
    *sim = nullptr;
    g_last_exception = e.what();
    return PION_GPU_EDEVICE;
  }
}
int pion_host_sim_create(const pion_gpu_config *cfg, int device, void **sim)
{
  try {
    *sim = new pion_host::sim_control_gpu(*cfg, device);
    return 0;
  }
  catch (const std::exception &e) {
    *sim = nullptr;
    g_last_exception = e.what();
    return PION_GPU_EDEVICE;
  }
}
void pion_host_sim_destroy(void *s) { delete static_cast<pion_host::sim_control_gpu *>(s); }
void *pion_host_sim_handle(void *s) { return static_cast<pion_host::sim_control_gpu *>(s)->handle(); }
int pion_host_sim_init(void *s, const double *P, double simtime, double finishtime, double first_dt_limit)
{
  auto *c = static_cast<pion_host::sim_control_gpu *>(s);
  c->T.finishtime = finishtime;
  c->T.first_step_dt_limit = first_dt_limit;
  return c->Init(P, simtime);
}
// restart: step counter and the previous step (SimParams::timestep / last_dt, read back from a snapshot header);
// a fresh problem on a re-used object: (0, 1e100)
int pion_host_sim_set_time(void *s, int timestep, double last_dt)
{
  auto *c = static_cast<pion_host::sim_control_gpu *>(s);
  c->T.timestep = timestep;
  c->T.last_dt = last_dt;
  return 0;
}
int pion_host_sim_time_int(void *s, int nsteps, double *simtime, double *last_dt)
{
  auto *c = static_cast<pion_host::sim_control_gpu *>(s);
  try {
    int n = c->Time_Int(nsteps);
    *simtime = c->T.simtime;
    *last_dt = c->T.last_dt;
    return n;
  }
  catch (const std::exception &e) {
    g_last_exception = e.what();   // the reference prints the rep.error text before exit(1); keep it readable
    return -1;
  }
}
// text of the last exception a pion_host_* call swallowed (empty if none), plus the handle's own error
int pion_host_sim_last_error(void *s, char *buf, int len)
{
  if (!buf || len <= 0) return PION_GPU_EINVAL;
  std::string m = g_last_exception;
  if (s) {
    const std::string h = static_cast<pion_host::sim_control_gpu *>(s)->last_error();
    if (!h.empty()) m += (m.empty() ? "" : " | ") + h;
  }
  snprintf(buf, (size_t)len, "%s", m.c_str());
  return 0;
}
int pion_host_sim_download(void *s, int which, double *P)
{
  auto *c = static_cast<pion_host::sim_control_gpu *>(s);
  if (int rc = c->finish_halo()) return rc;
  return c->download(which, P);
}
int pion_host_sim_finish_halo(void *s) { return static_cast<pion_host::sim_control_gpu *>(s)->finish_halo(); }
// hand the sim a slab communicator (pion_host_comm_create); call before pion_host_sim_init
int pion_host_sim_set_comm(void *s, void *comm)
{
  return static_cast<pion_host::sim_control_gpu *>(s)->set_comm(static_cast<pion_host::slab_comm *>(comm));
}
// one step at a time (bench.py times K of them between barriers)
int pion_host_sim_step(void *s, double *dt)
{
  auto *c = static_cast<pion_host::sim_control_gpu *>(s);
  try {
    int err = c->calculate_timestep();
    if (err) {
      g_last_exception = "calculate_timestep: " + c->last_error();
      return err;
    }
    *dt = c->advance_time();
    return 0;
  }
  catch (const std::exception &e) {
    g_last_exception = e.what();
    return -1;
  }
}
}
