// pion_backend_gpu.cpp -- pion_backend bound to libpion_gpu.so (include/pion_gpu.h): the only backend of the product.
#include "pion_backend.h"

#include <hip/hip_runtime_api.h>

namespace {

int gpu_dt_begin(void *h)
{
  void *d = nullptr;
  if (int rc = pion_gpu_calc_dt_device(h, &d)) return rc;
  return pion_gpu_dt_request(h);
}

// staged halos: in place from / into the state arrays (pion_gpu_halo_spans), on the communication stream the
// library orders against the compute stream (pion_gpu_halo_begin / _end)
hipStream_t comm_stream_of(void *h)
{
  void *cs = pion_gpu_get_stream(h, 1);
  return (hipStream_t)(cs ? cs : pion_gpu_get_stream(h, 0));
}
int gpu_halo_to_host_begin(void *h, int which, double *lo, double *hi)
{
  pion_gpu_halo_spans_t sp;
  if (int rc = pion_gpu_halo_spans(h, which, &sp)) return rc;
  if (int rc = pion_gpu_halo_begin(h)) return rc;   // after the stage + boundary kernels that wrote the planes
  hipStream_t s = comm_stream_of(h);
  const size_t n = (size_t)sp.count_per_var, nb = n * sizeof(double);
  for (int v = 0; v < sp.nvar; v++) {
    const long o = (long)v * sp.var_stride;
    if (lo && hipMemcpyAsync(lo + (size_t)v * n, sp.send_lo + o, nb, hipMemcpyDeviceToHost, s) != hipSuccess)
      return PION_GPU_EDEVICE;
    if (hi && hipMemcpyAsync(hi + (size_t)v * n, sp.send_hi + o, nb, hipMemcpyDeviceToHost, s) != hipSuccess)
      return PION_GPU_EDEVICE;
  }
  return 0;
}
int gpu_halo_to_host_end(void *h)
{
  return hipStreamSynchronize(comm_stream_of(h)) == hipSuccess ? 0 : PION_GPU_EDEVICE;
}
int gpu_halo_from_host(void *h, int which, const double *lo, const double *hi)
{
  pion_gpu_halo_spans_t sp;
  if (int rc = pion_gpu_halo_spans(h, which, &sp)) return rc;
  hipStream_t s = comm_stream_of(h);
  const size_t n = (size_t)sp.count_per_var, nb = n * sizeof(double);
  for (int v = 0; v < sp.nvar; v++) {
    const long o = (long)v * sp.var_stride;
    if (lo && hipMemcpyAsync(sp.recv_lo + o, lo + (size_t)v * n, nb, hipMemcpyHostToDevice, s) != hipSuccess)
      return PION_GPU_EDEVICE;
    if (hi && hipMemcpyAsync(sp.recv_hi + o, hi + (size_t)v * n, nb, hipMemcpyHostToDevice, s) != hipSuccess)
      return PION_GPU_EDEVICE;
  }
  // the host buffers may be reused by the caller: the copies must have left them
  if (hipStreamSynchronize(s) != hipSuccess) return PION_GPU_EDEVICE;
  return pion_gpu_halo_end(h);   // the z-boundary part of the next stage waits for this point
}

const pion_backend k_gpu = {
    "libpion_gpu.so",
    pion_gpu_create,
    pion_gpu_destroy,
    pion_gpu_last_error,
    pion_gpu_upload,
    pion_gpu_download,
    pion_gpu_update_bcs,
    pion_gpu_stage,
    pion_gpu_stage_part,
    pion_gpu_set_glm_speeds,
    pion_gpu_calc_dt,
    gpu_dt_begin,
    pion_gpu_dt_wait,
    pion_gpu_halo_count,
    gpu_halo_to_host_begin,
    gpu_halo_to_host_end,
    gpu_halo_from_host,
};

}  // namespace

extern "C" const pion_backend *pion_backend_gpu(void) { return &k_gpu; }
