// sim_control_gpu.h -- C++ host adapter above the C-ABI (include/pion_gpu.h).
//
// Mirrors the reference's caller side of the hot path, same member names and argument meaning:
//   sim_control::Time_Int            source/sim_control/sim_control.cpp:202-281
//   calc_timestep::calculate_timestep, timestep_checking_and_limiting
//                                     source/sim_control/calc_timestep.cpp:68-153, 219-262
//   time_integrator::advance_time, first_order_update, second_order_update
//                                     source/sim_control/time_integrator.cpp:72-250
// so that a maintainer can swap these three functions in a PION build and
// sim_control drives the GPU path unchanged.  Errors follow the reference's convention:
// int error counts are returned and accumulated; unrecoverable conditions throw (the reference
// calls rep.error -> exit(1)).
#ifndef PION_SIM_CONTROL_GPU_H
#define PION_SIM_CONTROL_GPU_H

#include <string>

#include "../../include/pion_gpu.h"
#include "pion_backend.h"

namespace pion_host {

class slab_comm;   // slab_comm.h: slab_comm_rccl (RCCL over xGMI) or slab_comm_shm (host-staged)

// the slice of SimParams (sim_params.h:200-285) the time loop itself reads/writes
struct SimTime {
  double simtime = 0.0, finishtime = 1e300, dt = 0.0, last_dt = 1e100, min_timestep = 0.0;
  int timestep = 0;
  double first_step_dt_limit = -1.0;  // wind / jet limit of calc_dynamics_dt (calc_timestep.cpp:313-323); <0: none
};

class sim_control_gpu {
 public:
  // backend: what runs below the time loop; null = the product's only one, libpion_gpu.so (pion_backend_gpu())
  sim_control_gpu(const pion_gpu_config &cfg, int device, const pion_backend *backend = nullptr);
  ~sim_control_gpu();
  sim_control_gpu(const sim_control_gpu &) = delete;

  // sim_init::Init (sim_init.cpp:219-267): ReadData, Ph=P, assign + update boundaries
  int Init(const double *P_soa, double simtime);
  // calc_timestep::calculate_timestep
  int calculate_timestep();
  // time_integrator::advance_time and its two stages
  double advance_time();
  int first_order_update(double dt, int ooa);
  int second_order_update(double dt, int ooa);
  // sim_control::Time_Int without I/O; nsteps<0: until finishtime
  int Time_Int(int nsteps);

  // z-slab of a larger domain: exchange the z ghost planes after every boundary update (under the
  // interior part of the next stage) and min-reduce the time step over the ranks
  // (sim_control_pllel, sim_control_MPI.cpp:482-583; MCMD_boundaries.cpp:122-237)
  int set_comm(slab_comm *c);
  int update_boundaries(int cstep, int maxstep, int assign);
  int stage(double dt, int space_ooa, int is_full);
  int finish_halo();
  int request_next_dt();

  int download(int which, double *P_soa) { return be_->download(h_, which, P_soa); }
  const pion_backend *backend() const { return be_; }
  void *handle() { return h_; }
  std::string last_error() const;

  SimTime T;
  pion_gpu_config cfg;

 private:
  const pion_backend *be_;
  void *h_;
  slab_comm *comm_ = nullptr;
  bool dt_requested_ = false;
};

}  // namespace pion_host
#endif
