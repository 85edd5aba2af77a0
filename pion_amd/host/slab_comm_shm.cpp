// slab_comm_shm.cpp -- see slab_comm_shm.h
#include "slab_comm_shm.h"

#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace pion_host {

// Segment layout (all ranks compute the same offsets):
//   [rank r][side 0 = from the lower neighbour, 1 = from the upper neighbour]  Box { header 64 B, count_ doubles }
//   [rank r] Red { header 64 B }    -- two alternating slots per rank for the min-reduction
struct slab_comm_shm::Box {
  std::atomic<unsigned long long> written;    // exchanges whose planes are complete in `data`
  std::atomic<unsigned long long> consumed;   // exchanges the owner has copied out
  char pad[64 - 2 * sizeof(std::atomic<unsigned long long>)];
  double data[1];
};
namespace {
struct Red {
  std::atomic<unsigned long long> seq[2];
  double v[2][2];
  char pad[64 - 16];
};
static_assert(sizeof(std::atomic<unsigned long long>) == 8, "lock-free 64-bit atomics expected");
double now_s()
{
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
}  // namespace

slab_comm_shm::slab_comm_shm(int rank, int world, bool periodic_z, const char *name, const pion_backend *backend)
    : rank_(rank), world_(world), up_(-1), down_(-1), name_(name ? name : ""), be_(backend ? backend : pion_backend_gpu()),
      h_(nullptr), count_(0), box_bytes_(0), seg_bytes_(0), seg_(nullptr), stage_lo_(nullptr), stage_hi_(nullptr),
      pinned_(false), pending_(-1), seq_(0), red_seq_(0), requested_(false)
{
  if (world < 1 || rank < 0 || rank >= world || name_.empty() || name_[0] != '/')
    throw std::runtime_error("slab_comm_shm: bad rank / world / segment name (\"/name\")");
  // decomposeDomain along z (MCMD_control.cpp:231-309): rank r is below r+1; periodic wrap 0 <-> world-1
  if (periodic_z || rank < world - 1) up_ = (rank + 1) % world;
  if (periodic_z || rank > 0) down_ = (rank - 1 + world) % world;
  if (world == 1 && !periodic_z) up_ = down_ = -1;
}

slab_comm_shm::~slab_comm_shm()
{
  if (seg_) munmap(seg_, seg_bytes_);
  if (rank_ == 0 && !name_.empty()) shm_unlink(name_.c_str());
  if (pinned_) {
    (void)hipHostFree(stage_lo_);
    (void)hipHostFree(stage_hi_);
  }
  else {
    free(stage_lo_);
    free(stage_hi_);
  }
}

slab_comm_shm::Box *slab_comm_shm::box(int rank, int side) const
{
  return reinterpret_cast<Box *>(seg_ + ((size_t)rank * 2 + side) * box_bytes_);
}

int slab_comm_shm::attach(void *handle)
{
  h_ = handle;
  count_ = be_->halo_count(h_);
  if (count_ <= 0) {
    err_ = "halo_count: not a 3-D slab";
    return PION_GPU_EINVAL;
  }
  box_bytes_ = ((64 + (size_t)count_ * sizeof(double)) + 63) / 64 * 64;
  seg_bytes_ = (size_t)world_ * 2 * box_bytes_ + (size_t)world_ * sizeof(Red);
  // every rank creates-or-opens and sizes the segment (same size everywhere; a fresh segment is zero-filled)
  const int fd = shm_open(name_.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)seg_bytes_) != 0) {
    err_ = "shm_open / ftruncate failed for " + name_;
    if (fd >= 0) close(fd);
    return PION_GPU_EDEVICE;
  }
  void *m = mmap(nullptr, seg_bytes_, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) {
    err_ = "mmap failed";
    return PION_GPU_EDEVICE;
  }
  seg_ = static_cast<char *>(m);
  // staging buffers: pinned when a HIP device is there (asynchronous copies), plain memory otherwise
  const size_t nb = (size_t)count_ * sizeof(double);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 && hipHostMalloc((void **)&stage_lo_, nb, hipHostMallocDefault) == hipSuccess
      && hipHostMalloc((void **)&stage_hi_, nb, hipHostMallocDefault) == hipSuccess)
    pinned_ = true;
  else {
    (void)hipGetLastError();
    stage_lo_ = static_cast<double *>(malloc(nb));
    stage_hi_ = static_cast<double *>(malloc(nb));
    if (!stage_lo_ || !stage_hi_) {
      err_ = "out of memory";
      return PION_GPU_EDEVICE;
    }
  }
  return 0;
}

// spin (yielding) until the 64-bit counter reaches `want`; a peer that died must not hang the run for ever
int slab_comm_shm::wait_until(const volatile void *counter, unsigned long long want, const char *what)
{
  const std::atomic<unsigned long long> *c = static_cast<const std::atomic<unsigned long long> *>(const_cast<const void *>(counter));
  const double t0 = now_s();
  unsigned spins = 0;
  while (c->load(std::memory_order_acquire) < want) {
    if (++spins > 64) sched_yield();
    if ((spins & 0xfff) == 0 && now_s() - t0 > 120.0) {
      err_ = std::string("slab_comm_shm: timed out waiting for ") + what;
      return PION_GPU_EDEVICE;
    }
  }
  return 0;
}

// first half: my top planes -> staging (hi), my bottom planes -> staging (lo), asynchronously
int slab_comm_shm::start(int which)
{
  if (pending_ >= 0) {
    err_ = "previous halo exchange not finished";
    return PION_GPU_EINVAL;
  }
  if (up_ < 0 && down_ < 0) return 0;
  if (!seg_) {
    err_ = "attach() first";
    return PION_GPU_EINVAL;
  }
  if (int rc = be_->halo_to_host_begin(h_, which, down_ >= 0 ? stage_lo_ : nullptr, up_ >= 0 ? stage_hi_ : nullptr)) return rc;
  pending_ = which;
  return 0;
}

// second half: (the host has meanwhile enqueued the interior part of the next stage) publish my planes in the
// neighbours' mailboxes, take theirs out of mine, put them into the ghost planes
int slab_comm_shm::finish()
{
  if (pending_ < 0) return 0;
  const int which = pending_;
  pending_ = -1;
  if (int rc = be_->halo_to_host_end(h_)) return rc;
  const size_t nb = (size_t)count_ * sizeof(double);
  const unsigned long long n = seq_ + 1;
  // my top planes are the upper neighbour's ZN ghosts: its "from the lower neighbour" box; the previous message
  // there must have been taken out
  if (up_ >= 0) {
    Box *b = box(up_, 0);
    if (int rc = wait_until(&b->consumed, seq_, "the upper neighbour to take the previous planes")) return rc;
    memcpy(b->data, stage_hi_, nb);
    b->written.store(n, std::memory_order_release);
  }
  if (down_ >= 0) {
    Box *b = box(down_, 1);
    if (int rc = wait_until(&b->consumed, seq_, "the lower neighbour to take the previous planes")) return rc;
    memcpy(b->data, stage_lo_, nb);
    b->written.store(n, std::memory_order_release);
  }
  // mine: from the lower neighbour -> ZN ghosts (staging lo), from the upper -> ZP ghosts (staging hi)
  if (down_ >= 0) {
    Box *b = box(rank_, 0);
    if (int rc = wait_until(&b->written, n, "the lower neighbour's planes")) return rc;
    memcpy(stage_lo_, b->data, nb);
    b->consumed.store(n, std::memory_order_release);
  }
  if (up_ >= 0) {
    Box *b = box(rank_, 1);
    if (int rc = wait_until(&b->written, n, "the upper neighbour's planes")) return rc;
    memcpy(stage_hi_, b->data, nb);
    b->consumed.store(n, std::memory_order_release);
  }
  seq_ = n;
  return be_->halo_from_host(h_, which, down_ >= 0 ? stage_lo_ : nullptr, up_ >= 0 ? stage_hi_ : nullptr);
}

int slab_comm_shm::request_min()
{
  const int rc = be_->dt_begin(h_);
  requested_ = (rc == 0);
  return rc;
}

// COMM->global_operation_double("MIN", .) over the ranks (comm_mpi.cpp:182-209): every rank publishes its pair in
// its slot of this round (two slots alternate: a rank can be at most one round ahead of the slowest) and reads all
int slab_comm_shm::allreduce_min(double *t_dyn, double *t_mp)
{
  if (!requested_) {
    if (int rc = request_min()) return rc;
  }
  requested_ = false;
  double a = 0.0, b = 0.0;
  if (int rc = be_->dt_wait(h_, &a, &b)) return rc;
  if (world_ > 1) {
    if (!seg_) {
      err_ = "attach() first";
      return PION_GPU_EINVAL;
    }
    Red *red = reinterpret_cast<Red *>(seg_ + (size_t)world_ * 2 * box_bytes_);
    const unsigned long long n = red_seq_ + 1;
    const int slot = (int)(n & 1);
    red[rank_].v[slot][0] = a;
    red[rank_].v[slot][1] = b;
    red[rank_].seq[slot].store(n, std::memory_order_release);
    for (int r = 0; r < world_; r++) {
      if (int rc = wait_until(&red[r].seq[slot], n, "a rank's time step")) return rc;
      a = (red[r].v[slot][0] < a) ? red[r].v[slot][0] : a;
      b = (red[r].v[slot][1] < b) ? red[r].v[slot][1] : b;
    }
    red_seq_ = n;
  }
  *t_dyn = a;
  *t_mp = b;
  return 0;
}

int slab_comm_shm::reset()
{
  requested_ = false;
  return finish();
}

}  // namespace pion_host

extern "C" {
// backend: null = libpion_gpu.so.  Returns a pion_host::slab_comm* (pion_host_sim_set_comm, pion_host_comm_destroy).
int pion_host_comm_shm_create(int rank, int world, int periodic_z, const char *name, const pion_backend *backend, void **comm)
{
  try {
    *comm = static_cast<pion_host::slab_comm *>(new pion_host::slab_comm_shm(rank, world, periodic_z != 0, name, backend));
    return 0;
  }
  catch (const std::exception &e) {
    fprintf(stderr, "pion_host_comm_shm_create: %s\n", e.what());
    *comm = nullptr;
    return PION_GPU_EINVAL;
  }
}
}
