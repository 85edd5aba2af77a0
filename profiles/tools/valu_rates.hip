// Issue cost of the VALU instructions the stage kernels are made of, relative to v_fma_f64, on gfx950:
//   hipcc --offload-arch=gfx950 -O2 profiles/tools/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
// Each kernel runs ITER x 64 independent instances of one instruction per wavefront (8 independent register
// chains, so latency is hidden), at 1 and at 2 wavefronts per SIMD; the result is SIMD cycles per instruction
// taken from s_memtime (100 MHz constant clock) and the measured v_fma_f64 rate (4 cycles per wave64 op).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY(NAME, ASM)                                                                                      \
  __global__ __launch_bounds__(256) void k_##NAME(double *out, int iters, double seed)                      \
  {                                                                                                          \
    double r0 = seed + threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
    double a = 1.0000001, b = 1e-9;                                                                          \
    for (int i = 0; i < iters; i++) {                                                                        \
      _Pragma("unroll") for (int u = 0; u < 8; u++) {                                                        \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                                 \
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)      \
                     : "v"(a), "v"(b) : "vcc");                                                              \
      }                                                                                                      \
    }                                                                                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                      \
  }

#define A_FMA(n) "v_fma_f64 %" #n ", %" #n ", %8, %9\n"
#define A_MUL(n) "v_mul_f64 %" #n ", %" #n ", %8\n"
#define A_ADD(n) "v_add_f64 %" #n ", %" #n ", %9\n"
#define A_RCP(n) "v_rcp_f64 %" #n ", %" #n "\n"
#define A_RSQ(n) "v_rsq_f64 %" #n ", %" #n "\n"
#define A_SQRT(n) "v_sqrt_f64 %" #n ", %" #n "\n"
#define A_MAX(n) "v_max_f64 %" #n ", %" #n ", %8\n"
#define A_MOV64(n) "v_mov_b64 %" #n ", %8\n"
#define A_CMP(n) "v_cmp_lt_f64 vcc, %" #n ", %8\n"
#define A_CND(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define A_LSHLADD(n) "v_lshl_add_u64 %" #n ", %" #n ", 0, %8\n"
#define A_MOV32(n) "v_mov_b32 %" #n ", %8\n"
#define A_BFI(n) "v_bfi_b32 %" #n ", %8, %9, %" #n "\n"
#define A_CLASS(n) "v_cmp_class_f64 vcc, %" #n ", %8\n"
#define A_DPP(n) "v_mov_b32_dpp %" #n ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_WSHR(n) "v_mov_b32_dpp %" #n ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n"

// 32-bit forms need 32-bit operands: use the low halves
#define BODY32(NAME, ASM)                                                                                    \
  __global__ __launch_bounds__(256) void k_##NAME(double *out, int iters, double seed)                      \
  {                                                                                                          \
    unsigned r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
    unsigned a = 12345u + (unsigned)seed, b = 777u;                                                          \
    for (int i = 0; i < iters; i++) {                                                                        \
      _Pragma("unroll") for (int u = 0; u < 8; u++) {                                                        \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                                 \
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)      \
                     : "v"(a), "v"(b) : "vcc", "s20", "s21");                                                \
      }                                                                                                      \
    }                                                                                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7);            \
  }

BODY(fma, A_FMA) BODY(mul, A_MUL) BODY(add, A_ADD) BODY(rcp, A_RCP) BODY(rsq, A_RSQ) BODY(sqrt, A_SQRT)
BODY(max, A_MAX) BODY(mov64, A_MOV64) BODY(cmp, A_CMP) BODY(lshladd, A_LSHLADD)
#define A_CND64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]\n"
#define A_CNDI(n) "v_cndmask_b32 %" #n ", %8, %9, vcc\n"
#define A_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define A_ADD32(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define A_FMA32(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define A_CND64V(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, vcc\n"
#define A_CMPCND(n) "v_cmp_lt_u32 vcc, %" #n ", %8\nv_cndmask_b32 %" #n ", %" #n ", %8, vcc\nv_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
#define A_CMPCNDS(n) "v_cmp_lt_u32_e64 s[20:21], %" #n ", %8\nv_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]\nv_cndmask_b32_e64 %" #n ", %" #n ", %9, s[20:21]\n"
#define A_CNDMIX(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\nv_add_u32 %" #n ", %" #n ", %8\nv_add_u32 %" #n ", %" #n ", %9\nv_add_u32 %" #n ", %" #n ", %8\n"
BODY32(cnd64v, A_CND64V) BODY32(cmpcnd, A_CMPCND) BODY32(cmpcnds, A_CMPCNDS) BODY32(cndmix, A_CNDMIX)
BODY32(cnd, A_CND) BODY32(cnd64, A_CND64) BODY32(cndi, A_CNDI) BODY32(and32, A_AND) BODY32(add32, A_ADD32) BODY32(fma32, A_FMA32) BODY32(mov32, A_MOV32) BODY32(bfi, A_BFI) BODY32(dpp, A_DPP)

typedef void (*kern_t)(double *, int, double);
struct Ent { const char *name; kern_t k; };

int main()
{
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
  const int ncu = p.multiProcessorCount;
  double *out;
  hipMalloc(&out, sizeof(double) * 256 * ncu * 8);
  std::vector<Ent> ks = {{"v_fma_f64", k_fma}, {"v_mul_f64", k_mul}, {"v_add_f64", k_add}, {"v_max_f64", k_max},
                         {"v_rcp_f64", k_rcp}, {"v_rsq_f64", k_rsq}, {"v_sqrt_f64", k_sqrt}, {"v_mov_b64", k_mov64},
                         {"v_cmp_lt_f64", k_cmp}, {"v_lshl_add_u64", k_lshladd},
                         {"v_cndmask_b32 (vcc, dst=src0)", k_cnd}, {"v_cndmask_b32_e64 (sgpr)", k_cnd64}, {"v_cndmask_b32 (vcc, dst only)", k_cndi},
                         {"v_cndmask_b32_e64 (vcc)", k_cnd64v}, {"cmp vcc + 2 cndmask e32   [x3]", k_cmpcnd}, {"cmp sgpr + 2 cndmask e64 [x3]", k_cmpcnds},
                         {"cndmask e32 + 3 add_u32   [x4]", k_cndmix}, {"v_and_b32", k_and32}, {"v_add_u32", k_add32}, {"v_fma_f32", k_fma32}, {"v_mov_b32", k_mov32}, {"v_bfi_b32", k_bfi}, {"v_mov_b32 dpp row_shr", k_dpp}};
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double fma_ms[3] = {0, 0, 0};
  for (int wps = 1; wps <= 2; wps++) {
    printf("-- %d wavefront(s) per SIMD (%d workgroups of 256 on %d CUs)\n", wps, ncu * wps, ncu);
    for (auto &e : ks) {
      hipLaunchKernelGGL(e.k, dim3(ncu * wps), dim3(256), 0, 0, out, 100, 1.0);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(e.k, dim3(ncu * wps), dim3(256), 0, 0, out, iters, 1.0);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double ninst = (double)iters * 64 * wps;   // per SIMD
      if (e.k == k_fma) fma_ms[wps] = ms;
      printf("%-32s %8.3f ms  %7.2f ns per instruction per SIMD  = %5.2f x v_fma_f64 (%.1f cycles if fma = 4)\n", e.name, ms,
             ms * 1e6 / ninst, ms / fma_ms[wps], 4.0 * ms / fma_ms[wps]);
    }
  }
  printf("v_fma_f64 at 4 cycles per wave64 instruction => clock %.3f GHz (1 wavefront per SIMD), %.3f GHz (2)\n",
         4.0 * iters * 64 / (fma_ms[1] * 1e6), 4.0 * iters * 64 * 2 / (fma_ms[2] * 1e6));
  return 0;
}
