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

// The stage kernel's x tiling on one variable plane: rows of NXA = 516 doubles (512 + 2 x 2 ghosts), one wavefront per
// 62-cell tile reading 64 consecutive doubles from x0 = 62 * tile + 1 (so that wavefront reads start off the 128-byte
// line grid and neighbouring tiles overlap by two cells), every row once.  Unique bytes touched = rows * 516 * 8:
// what FETCH_SIZE x correction should give if the correction measured on aligned streams also holds for this pattern.
__global__ __launch_bounds__(256) void k_calib_tiles(const double *in, double *out, long nrows)
{
  const int NXA = 516, NT = 9;   // 8 full tiles + the 16-cell remainder
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long row = w / NT;
  const int tile = (int)(w % NT), lane = threadIdx.x & 63;
  if (row >= nrows) return;
  int x = 62 * tile + 1 + lane;
  if (x > NXA - 1) x = NXA - 1;
  const double v = in[row * NXA + x];
  if (lane == 0 && v == 12345.678) out[0] = v;   // (keeps the load; never true for the zero-filled input)
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
  {
    const long nrows = (4L * n) / 516;   // the whole 4 GiB input as rows of 516 doubles
    const unsigned nbt = (unsigned)((nrows * 9 + 3) / 4);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_calib_tiles, dim3(nbt), dim3(256), 0, 0, in, out, nrows);
    printf("k_calib_tiles reads %ld unique B (rows of 516 doubles, 62-cell tiles of 64 lanes)\n", nrows * 516 * 8);
  }
  (void)hipDeviceSynchronize();
  printf("calib done: k_calib<4,1> reads %ld B writes %ld B; k_calib<1,2> reads %ld B writes %ld B\n",
         4 * n * 8, 1 * n * 8, 1 * n * 8, 2 * n * 8);
  (void)hipFree(in);
  (void)hipFree(out);
  return 0;
}
