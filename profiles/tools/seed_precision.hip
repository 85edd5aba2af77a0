// Measures the relative error of the gfx950 v_rcp_f64 / v_rsq_f64 seeds and of the refinement steps the
// fast build uses (dev_eqns.h: rcp_pos, sqrt_rsqrt_pos).  hipcc --offload-arch=gfx950 -O2 -o seed_precision
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *o, int n)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r = __builtin_amdgcn_rcp(v);
  o[i] = r;
  double e = __builtin_fma(-v, r, 1.0);
  r = __builtin_fma(r, e, r);
  o[n + i] = r;
  e = __builtin_fma(-v, r, 1.0);
  r = __builtin_fma(r, e, r);
  o[2 * n + i] = r;
  double y = __builtin_amdgcn_rsq(v);
  o[3 * n + i] = y;
  double g = v * y, h = 0.5 * y;
  double rr = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, rr, g);
  h = __builtin_fma(h, rr, h);
  o[4 * n + i] = g;
  double d = __builtin_fma(-g, g, v);
  o[5 * n + i] = __builtin_fma(d, h, g);
  const double root = o[5 * n + i], r0 = h + h;
  o[6 * n + i] = __builtin_fma(r0, __builtin_fma(-root, r0, 1.0), r0);
}
int main()
{
  const int n = 1 << 20;
  std::vector<double> x(n), o(7 * n);
  unsigned long long s = 88172645463325252ULL;
  for (int i = 0; i < n; i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (double)(s >> 11) / 9007199254740992.0;
    x[i] = std::exp((u - 0.5) * 120.0) * (1.0 + u);   // 1e-26 .. 1e26
  }
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 7 * n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(o.data(), dout, 7 * n * 8, hipMemcpyDeviceToHost);
  const char *nm[7] = {"rcp seed", "rcp + 1 NR", "rcp + 2 NR", "rsq seed", "sqrt after 1 Goldschmidt", "sqrt + residual", "rsqrt (2h + 1 NR)"};
  for (int j = 0; j < 7; j++) {
    double m = 0;
    for (int i = 0; i < n; i++) {
      long double ex = (j < 3) ? 1.0L / x[i] : ((j == 3 || j == 6) ? 1.0L / sqrtl((long double)x[i]) : sqrtl((long double)x[i]));
      double e = (double)fabsl(((long double)o[j * n + i] - ex) / ex);
      if (e > m) m = e;
    }
    printf("%-28s max rel err %.3e  (2^%.1f)\n", nm[j], m, std::log2(m > 0 ? m : 1e-300));
  }
  return 0;
}
