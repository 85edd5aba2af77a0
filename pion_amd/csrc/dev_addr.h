// dev_addr.h -- "uniform base + 32-bit per-lane offset" addressing helpers shared by k_stage_rows2 and the
// marching prepass (included by kernels_fp.hip inside namespace pion::PION_FPNS).
#ifndef PION_DEV_ADDR_H
#define PION_DEV_ADDR_H

// Addressing: every global access of the kernel is "uniform base + 32-bit per-lane byte offset"
// (global_load ... v_off, s[base:base+1]): the variable, the neighbour shift along y / z and the array are
// folded into the scalar base, the x neighbours into the instruction's immediate offset, and the one
// per-lane quantity -- the cell -- is a single VGPR per row.  (With 64-bit per-lane addresses the compiler
// hoists one VGPR pair per load site out of the task loop: ~80 registers, which do not exist here.)
// Needs 8 * ncell < 2^32 (pion_gpu_create picks the cell-per-thread kernel otherwise; 512^3 with ghosts is 1.1e9).
// (readfirstlane keeps the optimiser from re-associating base + offset into per-lane 64-bit arithmetic; on a
// value that already lives in SGPRs it costs nothing.  The access is made through an address_space(1)
// pointer so that it stays a global_ instruction after the integer round trip.)
PDEV unsigned long long uni(const void *p)
{
  const unsigned long long x = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)x);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(x >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
PDEV double ldu(const char *ubase, const unsigned off)
{
  typedef const __attribute__((address_space(1))) char *gc;
  typedef const __attribute__((address_space(1))) double *gp;
  return *(gp)((gc)uni(ubase) + off);
}
PDEV unsigned ldub(const char *ubase, const unsigned off)
{
  typedef const __attribute__((address_space(1))) unsigned char *gp;
  return *((gp)uni(ubase) + off);
}
PDEV void stub(char *ubase, const unsigned off, const unsigned char x)
{
  typedef __attribute__((address_space(1))) unsigned char *gp;
  *((gp)uni(ubase) + off) = x;
}
// an SGPR zero the optimiser cannot see through: added to an array base inside a task it keeps the
// (loop-invariant) scalar address arithmetic of that task from being hoisted out of the row / plane loops,
// where its ~70 base pairs would have to be spilled (SGPR spills cost VALU lane moves)
PDEV unsigned opaque_zero()
{
  unsigned z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  return z;
}
// The per-lane offset as the block that uses it sees it.  Instruction selection works one basic block at a time
// and folds "uniform base + zext(32-bit lane offset)" into the scalar-base form of a global access
// (global_load v, v_off, s[base:base+1]) only when the zero-extension is in the block of the access; the
// optimiser otherwise keeps ONE 64-bit copy of the offset per row (hoisted) and every access pays a 64-bit
// VALU add (v_lshl_add_u64) for its address.  An empty volatile asm re-defines the offset inside the block.
PDEV unsigned pin_v(unsigned x)
{
  asm volatile("" : "+v"(x));
  return x;
}
// Neighbour-lane exchange by DPP wave shifts (gfx9: v_mov_b32_dpp ... wave_shl:1 / wave_shr:1, two per double):
// lane i receives lane i+1's (lane_next) or lane i-1's (lane_prev) value.  Same data movement as
// __shfl_down(x, 1, 64) / __shfl_up(x, 1, 64), which the compiler turns into ds_bpermute_b32 (an LDS-pipeline
// round trip each, behind an s_waitcnt lgkmcnt) -- here it is a 32-bit VALU move with no wait.  The lane at the
// end of the wavefront keeps its own value (bound_ctrl off), as __shfl_* does.
#ifndef PION_DPP_SHIFT
#define PION_DPP_SHIFT 1
#endif
PDEV double lane_next(const double x)
{
#if PION_DPP_SHIFT
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);   // wave_shl:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
#else
  return __shfl_down(x, 1, 64);
#endif
}
PDEV double lane_prev(const double x)
{
#if PION_DPP_SHIFT
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);   // wave_shr:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
#else
  return __shfl_up(x, 1, 64);
#endif
}
#endif
