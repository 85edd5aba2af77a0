// slab_comm_shm.h -- the slab protocol of slab_comm.h through host memory shared by the ranks of one node.
//
// Same role as slab_comm_rccl (it replaces comm_mpi::send_cell_data / receive_cell_data, comms/comm_mpi.cpp:287-425,
// and global_operation_double("MIN"), :182-209, for this path), different wire: the planes of a face leave the
// device into a pinned staging buffer (pion_backend::halo_to_host_begin, on the communication stream, while the
// compute stream runs the interior part of the next stage), are copied into the neighbour's mailbox in a POSIX
// shared-memory segment, and enter the neighbour's ghost planes from there (halo_from_host).  The time-step
// minimum is reduced through the same segment.  No MPI, no RCCL, no torch: ranks need only a common segment name.
// Use: ranks that share one GPU (tests), boxes without an RCCL peer, fall-back transport of bench.py.
#ifndef PION_SLAB_COMM_SHM_H
#define PION_SLAB_COMM_SHM_H

#include <string>

#include "pion_backend.h"
#include "slab_comm.h"

namespace pion_host {

class slab_comm_shm : public slab_comm {
 public:
  // name: POSIX shared-memory name common to the ranks of this run ("/pion_<launcher pid>"); every rank calls
  // with the same world / periodic_z.  backend: null = libpion_gpu.so.
  slab_comm_shm(int rank, int world, bool periodic_z, const char *name, const pion_backend *backend = nullptr);
  ~slab_comm_shm() override;
  slab_comm_shm(const slab_comm_shm &) = delete;

  int attach(void *handle) override;
  int start(int which) override;
  int finish() override;
  int request_min() override;
  int allreduce_min(double *t_dyn, double *t_mp) override;
  int reset() override;
  const std::string &last_error() const override { return err_; }

 private:
  struct Box;   // one mailbox (header + planes) inside the segment
  Box *box(int rank, int side) const;
  int wait_until(const volatile void *counter, unsigned long long want, const char *what);

  int rank_, world_, up_, down_;
  std::string name_;
  const pion_backend *be_;
  void *h_;
  long count_;                 // doubles per face
  size_t box_bytes_, seg_bytes_;
  char *seg_;                  // the mapped segment
  double *stage_lo_, *stage_hi_;   // pinned staging: my bottom / top on-grid planes (out), then the ghosts (in)
  bool pinned_;
  int pending_;                // array of the exchange in flight, or -1
  unsigned long long seq_;     // exchanges completed
  unsigned long long red_seq_; // reductions completed
  bool requested_;
  std::string err_;
};

}  // namespace pion_host
#endif
