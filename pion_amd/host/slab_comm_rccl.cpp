// slab_comm_rccl.cpp -- see slab_comm_rccl.h
#include "slab_comm_rccl.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace pion_host {

#define PH_HIP(call)                                                                 \
  do {                                                                               \
    hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess) {                                                          \
      err_ = std::string(#call) + ": " + hipGetErrorString(e_);                      \
      return PION_GPU_EDEVICE;                                                       \
    }                                                                                \
  } while (0)
#define PH_NCCL(call)                                                                \
  do {                                                                               \
    ncclResult_t r_ = (call);                                                        \
    if (r_ != ncclSuccess) {                                                         \
      err_ = std::string(#call) + ": " + ncclGetErrorString(r_);                     \
      return PION_GPU_EDEVICE;                                                       \
    }                                                                                \
  } while (0)

int slab_comm_rccl::get_unique_id(void *out128)
{
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return PION_GPU_EDEVICE;
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  memcpy(out128, &id, sizeof id);
  return 0;
}

slab_comm_rccl::slab_comm_rccl(int rank, int world, bool periodic_z, const void *unique_id, int device)
    : rank_(rank), world_(world), device_(device), up_(-1), down_(-1), comm_(nullptr), cstream_(nullptr), h_(nullptr),
      pending_(-1), requested_(false)
{
  // decomposeDomain along z (MCMD_control.cpp:231-309): rank r is below r+1; periodic wrap 0 <-> world-1
  if (periodic_z || rank < world - 1) up_ = (rank + 1) % world;
  if (periodic_z || rank > 0) down_ = (rank - 1 + world) % world;
  if (world == 1 && !periodic_z) up_ = down_ = -1;
  if (hipSetDevice(device) != hipSuccess) throw std::runtime_error("slab_comm_rccl: hipSetDevice failed");
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof id);
  ncclComm_t c = nullptr;
  const ncclResult_t r = ncclCommInitRank(&c, world, id, rank);
  if (r != ncclSuccess) throw std::runtime_error(std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  comm_ = c;
}

slab_comm_rccl::~slab_comm_rccl()
{
  (void)hipSetDevice(device_);
  if (cstream_) {
    (void)hipStreamSynchronize((hipStream_t)cstream_);
    if (h_) (void)pion_gpu_set_comm_stream(h_, nullptr);
  }
  if (comm_) (void)ncclCommDestroy((ncclComm_t)comm_);
  if (cstream_) (void)hipStreamDestroy((hipStream_t)cstream_);
}

int slab_comm_rccl::attach(void *gpu_handle)
{
  h_ = gpu_handle;
  PH_HIP(hipSetDevice(device_));
  // a high-priority stream: the exchange must not queue behind the interior part of the stage
  int lo = 0, hi = 0;
  PH_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t s;
  PH_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
  cstream_ = s;
  if (int rc = pion_gpu_synchronize(h_)) return rc;
  return pion_gpu_set_comm_stream(h_, cstream_);
}

// The nbc planes next to a z face are one contiguous run per variable of the SoA state, so they are sent
// from and received into the state array itself: no pack / unpack kernels, no staging buffers.  While the
// transfer is in flight the compute stream runs the INTERIOR part of the next stage, which reads no z ghost
// plane and writes the other array.
int slab_comm_rccl::start(int which)
{
  if (pending_ >= 0) {
    err_ = "previous halo exchange not finished";
    return PION_GPU_EINVAL;
  }
  if (up_ < 0 && down_ < 0) return 0;
  if (!h_) {
    err_ = "attach() first";
    return PION_GPU_EINVAL;
  }
  PH_HIP(hipSetDevice(device_));
  pion_gpu_halo_spans_t sp;
  int rc = pion_gpu_halo_spans(h_, which, &sp);
  if (rc) return rc;
  if ((rc = pion_gpu_halo_begin(h_))) return rc;   // after the stage + boundary kernels that wrote the planes
  hipStream_t s = (hipStream_t)cstream_;
  ncclComm_t c = (ncclComm_t)comm_;
  const size_t n = (size_t)sp.count_per_var;
  // one group; per variable: my top planes -> the upper neighbour's ZN ghosts, my bottom planes -> the lower
  // neighbour's ZP ghosts.  With two ranks and periodic z (or a rank that is its own neighbour) every message
  // goes to the same peer and is matched in the order posted, which is the same on both sides.
  PH_NCCL(ncclGroupStart());
  for (int v = 0; v < sp.nvar; v++) {
    const long o = (long)v * sp.var_stride;
    if (up_ >= 0) PH_NCCL(ncclSend(sp.send_hi + o, n, ncclDouble, up_, c, s));
    if (down_ >= 0) PH_NCCL(ncclRecv(sp.recv_lo + o, n, ncclDouble, down_, c, s));
    if (down_ >= 0) PH_NCCL(ncclSend(sp.send_lo + o, n, ncclDouble, down_, c, s));
    if (up_ >= 0) PH_NCCL(ncclRecv(sp.recv_hi + o, n, ncclDouble, up_, c, s));
  }
  PH_NCCL(ncclGroupEnd());
  pending_ = which;
  return 0;
}

int slab_comm_rccl::finish()
{
  if (pending_ < 0) return 0;
  pending_ = -1;
  PH_HIP(hipSetDevice(device_));
  // the ghost planes are complete once the communication stream has passed this point; the host does not wait
  return pion_gpu_halo_end(h_);
}

// enqueue: reduction of the device-resident {t_dyn, t_mp} over the ranks + copy to pinned host memory
int slab_comm_rccl::request_min()
{
  void *d = nullptr;
  if (int rc = pion_gpu_calc_dt_device(h_, &d)) return rc;
  if (world_ > 1) {
    PH_HIP(hipSetDevice(device_));
    // in place, on the compute stream, behind the reduction kernel / the stage that left the minima
    PH_NCCL(ncclAllReduce(d, d, 2, ncclDouble, ncclMin, (ncclComm_t)comm_, (hipStream_t)pion_gpu_get_stream(h_, 0)));
  }
  const int rc = pion_gpu_dt_request(h_);
  requested_ = (rc == 0);
  return rc;
}

int slab_comm_rccl::allreduce_min(double *t_dyn, double *t_mp)
{
  if (!requested_) {
    if (int rc = request_min()) return rc;
  }
  requested_ = false;
  return pion_gpu_dt_wait(h_, t_dyn, t_mp);
}

int slab_comm_rccl::reset()
{
  requested_ = false;
  return finish();
}

}  // namespace pion_host

// ---- C view (ctypes: tests, bench) ------------------------------------------------------------
extern "C" {
int pion_host_comm_unique_id(void *out128) { return pion_host::slab_comm_rccl::get_unique_id(out128); }
int pion_host_comm_create(int rank, int world, int periodic_z, const void *unique_id, int device, void **comm)
{
  try {
    *comm = static_cast<pion_host::slab_comm *>(new pion_host::slab_comm_rccl(rank, world, periodic_z != 0, unique_id, device));
    return 0;
  }
  catch (const std::exception &e) {
    fprintf(stderr, "pion_host_comm_create: %s\n", e.what());
    *comm = nullptr;
    return PION_GPU_EDEVICE;
  }
}
void pion_host_comm_destroy(void *c) { delete static_cast<pion_host::slab_comm *>(c); }
int pion_host_comm_attach(void *c, void *gpu_handle) { return static_cast<pion_host::slab_comm *>(c)->attach(gpu_handle); }
int pion_host_comm_start(void *c, int which) { return static_cast<pion_host::slab_comm *>(c)->start(which); }
int pion_host_comm_finish(void *c) { return static_cast<pion_host::slab_comm *>(c)->finish(); }
int pion_host_comm_allreduce_min(void *c, double *t_dyn, double *t_mp)
{
  return static_cast<pion_host::slab_comm *>(c)->allreduce_min(t_dyn, t_mp);
}
int pion_host_comm_last_error(void *c, char *buf, int len)
{
  snprintf(buf, (size_t)len, "%s", static_cast<pion_host::slab_comm *>(c)->last_error().c_str());
  return 0;
}
}
