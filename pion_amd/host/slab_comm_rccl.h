// slab_comm_rccl.h -- z-slab halo exchange and time-step reduction over RCCL, driven from C++.
//
// Replaces, for the flux-update path on one node (one process per GPU), the reference's
//   comm_mpi::send_cell_data / receive_cell_data        source/comms/comm_mpi.cpp:287-425
//   comm_mpi::global_operation_double("MIN", .)         source/comms/comm_mpi.cpp:182-209
//   MCMD_bc::BC_update_BCMPI                            source/boundaries/MCMD_boundaries.cpp:122-237
//   sim_control_pllel::calculate_timestep (global min)  source/sim_control/sim_control_MPI.cpp:482-583
// with what SURVEY.md s5 specifies: per stage ONE ncclGroupStart / ncclSend x2 / ncclRecv x2 /
// ncclGroupEnd on a communication stream (point-to-point over xGMI, two peers per GPU), and ONE
// ncclAllReduce(ncclMin) over the two doubles {t_dyn, t_mp} that the stage kernel left on the device;
// the only host synchronisation per step is the 16-byte read-back of the reduced minima.
//
// The planes travel in place (pion_gpu_halo_spans: one contiguous run per variable and face, sent from and
// received into the state arrays, no pack / unpack kernels); the stream ordering lives behind the C-ABI
// (pion_gpu_halo_begin / _end, pion_gpu_set_comm_stream, pion_gpu_stage_part).  This class owns the RCCL
// communicator and the communication stream, nothing else.
#ifndef PION_SLAB_COMM_RCCL_H
#define PION_SLAB_COMM_RCCL_H

#include <string>

#include "../../include/pion_gpu.h"
#include "slab_comm.h"

namespace pion_host {

class slab_comm_rccl : public slab_comm {
 public:
  // unique_id: the 128 bytes of an ncclUniqueId made by rank 0 (get_unique_id) and handed to every
  // rank by the launcher.  periodic_z: the global z faces are periodic (rank 0 <-> world-1 exchange).
  // world == 1 with periodic_z is the loop-back case: the rank is its own neighbour (RCCL send/recv to
  // self), which reproduces the single-domain periodic run bit for bit.
  slab_comm_rccl(int rank, int world, bool periodic_z, const void *unique_id, int device);
  ~slab_comm_rccl() override;
  slab_comm_rccl(const slab_comm_rccl &) = delete;

  static int get_unique_id(void *out128);

  // create the communication stream and register it (pion_gpu_set_comm_stream); must precede start()
  int attach(void *gpu_handle) override;
  // BC_update_BCMPI, first half: pack the on-grid planes next to the internal z faces of array `which`
  // (0 = P, 1 = Ph) and enqueue the grouped send / recv.  Returns at once.
  int start(int which) override;
  // second half: unpack into the ghost planes (communication stream; the library orders the
  // z-boundary part of the next stage after it)
  int finish() override;
  // global minimum of the device-resident {t_dyn, t_mp}: request_min() enqueues the all-reduce and the copy
  // to pinned host memory and returns (call it right after the full-step stage); allreduce_min() waits for
  // it (requesting first if nobody has) -- the single host synchronisation of a step
  int request_min() override;
  int allreduce_min(double *t_dyn, double *t_mp) override;
  // new state uploaded (sim_control_gpu::Init): complete an exchange in flight, forget a pending request_min()
  int reset() override;

  bool has_neighbours() const { return up_ >= 0 || down_ >= 0; }
  const std::string &last_error() const override { return err_; }

 private:
  int rank_, world_, device_, up_, down_;
  void *comm_;      // ncclComm_t
  void *cstream_;   // hipStream_t
  void *h_;         // pion_gpu handle
  int pending_;     // array of the exchange in flight, or -1
  bool requested_;  // request_min() issued, allreduce_min() not yet
  std::string err_;
};

}  // namespace pion_host
#endif
