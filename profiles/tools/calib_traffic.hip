// calib_traffic.hip -- known-byte-count kernels with the stage kernel's access width (one 8-byte
// global_load_dwordx2 / global_store_dwordx2 per lane, SoA planes), used to calibrate rocprofv3's
// FETCH_SIZE / WRITE_SIZE on gfx950 as MI355X_MICROARCH.md (HBM section) asks for access widths
// other than 16 B/lane.  Build: hipcc --offload-arch=gfx950 -O2 -o calib_traffic calib_traffic.hip
// Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace`.
#include <hip/hip_runtime.h>
#include <cstdio>

// reads NREAD planes of n doubles, writes NWRITE planes (n far beyond the 256 MiB Infinity Cache)
template <int NREAD, int NWRITE>
__global__ __launch_bounds__(256) void k_calib(const double *in, double *out, long n)
{
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
#pragma unroll
  for (int v = 0; v < NREAD; v++) s += in[v * n + i];
#pragma unroll
  for (int v = 0; v < NWRITE; v++) out[v * n + i] = s + v;
}

int main()
{
  const long n = 1L << 27;  // 1 GiB per plane
  double *in, *out;
  if (hipMalloc(&in, sizeof(double) * n * 4) != hipSuccess) return 1;
  if (hipMalloc(&out, sizeof(double) * n * 2) != hipSuccess) return 1;
  (void)hipMemset(in, 0, sizeof(double) * n * 4);
  (void)hipMemset(out, 0, sizeof(double) * n * 2);
  const unsigned nb = (unsigned)((n + 255) / 256);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL((k_calib<4, 1>), dim3(nb), dim3(256), 0, 0, in, out, n);  // 4 GiB read, 1 GiB written
    hipLaunchKernelGGL((k_calib<1, 2>), dim3(nb), dim3(256), 0, 0, in, out, n);  // 1 GiB read, 2 GiB written
  }
  (void)hipDeviceSynchronize();
  printf("calib done: k_calib<4,1> reads %ld B writes %ld B; k_calib<1,2> reads %ld B writes %ld B\n",
         4 * n * 8, 1 * n * 8, 1 * n * 8, 2 * n * 8);
  (void)hipFree(in);
  (void)hipFree(out);
  return 0;
}
