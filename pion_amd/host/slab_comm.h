// slab_comm.h -- what sim_control_gpu needs from a z-slab communicator (one process per GPU): the halo
// exchange of BC_update_BCMPI (boundaries/MCMD_boundaries.cpp:122-237) split into a start and a finish around the
// interior part of a stage, and the global minimum of the time step (sim_control_MPI.cpp:503-504).
//
// Two implementations:
//   slab_comm_rccl  (slab_comm_rccl.h)  device-resident transfers over RCCL / xGMI -- the production transport;
//   slab_comm_shm   (slab_comm_shm.h)   the same protocol through host memory shared by the ranks of one node
//                                       (pinned staging buffers, POSIX shared-memory mailboxes): for ranks that
//                                       share one GPU, for boxes without an RCCL peer, and as the fall-back
//                                       transport (`bench.py --transport shm`).
#ifndef PION_SLAB_COMM_H
#define PION_SLAB_COMM_H

#include <string>

namespace pion_host {

class slab_comm {
 public:
  virtual ~slab_comm() {}
  // bind to the backend handle whose state is exchanged; must precede start()
  virtual int attach(void *handle) = 0;
  // first half: the on-grid planes next to the internal z faces of array `which` (0 = P, 1 = Ph) leave.
  // Returns at once.
  virtual int start(int which) = 0;
  // second half: the ghost planes are (ordered to be) filled before the z-boundary part of the next stage
  virtual int finish() = 0;
  // global minimum of the device-resident {t_dyn, t_mp}: request_min() starts it (right after the full-step
  // stage), allreduce_min() waits for it -- the single host synchronisation of a step
  virtual int request_min() = 0;
  virtual int allreduce_min(double *t_dyn, double *t_mp) = 0;
  // a new state was uploaded (sim_control_gpu::Init): complete an exchange in flight, forget a pending request
  virtual int reset() = 0;
  virtual const std::string &last_error() const = 0;
};

}  // namespace pion_host
#endif
