// dev_eqns.h -- device-side equations for the gfx950 flux-update kernels.
//
// Everything here works on a per-thread state vector held in registers and
// expressed in the SWEEP FRAME: slot 2 is the velocity normal to the interface,
// slots 3,4 the two transverse components in the cyclic order the reference's
// index permutation produces (source/equations/eqns_base.cpp:94-132:
// X:(x,y,z)  Y:(y,z,x)  Z:(z,x,y)), likewise slots 5,6,7 for B.  The state is
// rotated when it is loaded from / stored to the SoA arrays (gvar<AXIS>()), so
// no index is ever computed at run time and every array below lives in VGPRs.
//
// The arithmetic -- operand order included -- is that of
//   source/equations/eqns_hydro_adiabatic.cpp:89-346   (eqns_Euler)
//   source/equations/eqns_mhd_adiabatic.cpp:79-278,598-660 (eqns_mhd_ideal, _mixedGLM)
//   source/spatial_solvers/solver_eqn_hydro_adi.cpp:211-273, solver_eqn_mhd_adi.cpp:288-366,846-904
//     (tracer extensions of PtoU/UtoP/PUtoFlux/UtoFlux)
// so that the strict build (-ffp-contract=off) reproduces the reference bit for bit.
#ifndef PION_DEV_EQNS_H
#define PION_DEV_EQNS_H

#include <hip/hip_runtime.h>

#define PDEV __device__ __forceinline__

namespace pion {

// constants.h:150-157,336-339
#define PION_SMALLVALUE 1.0e-12
#define PION_MACHINEACCURACY 5.e-16
#define PION_TINYVALUE 1.0e-100
#define PION_VERY_TINY_VALUE 1.0e-200
#define PION_BASEPG 1.e-5

enum { EQEUL = 1, EQMHD = 2, EQGLM = 3 };
// sweep-frame slots
enum { qRO = 0, qPG = 1, qVN = 2, qVT1 = 3, qVT2 = 4, qBN = 5, qBT1 = 6, qBT2 = 7, qSI = 8 };
// conserved names for the same slots
enum { uRHO = 0, uERG = 1, uMN = 2, uMT1 = 3, uMT2 = 4, uBN = 5, uBT1 = 6, uBT2 = 7, uPSI = 8 };

// error bits reported through the device error word
enum { ERR_NEG_DENSITY = 1, ERR_RIEMANN_INPUT = 2, ERR_COOLING = 4, ERR_BAD_DT = 8, ERR_MHD_RIEMANN = 16 };

// std::max / std::min semantics (not fmax/fmin: NaN behaviour differs)
PDEV double dmax(double a, double b) { return (a < b) ? b : a; }
PDEV double dmin(double a, double b) { return (b < a) ? b : a; }
#ifdef PION_FAST_MATH
// Fast build only.  One-instruction max/min (a NaN operand is dropped instead of, in one of the two
// argument orders, returned: only differs for states that are already in error).
PDEV double fmx(double a, double b) { return __builtin_fmax(a, b); }
PDEV double fmn(double a, double b) { return __builtin_fmin(a, b); }
// 1/x: v_rcp_f64 seed (measured 2^-24.4 on gfx950, profiles/tools/seed_precision.hip) and two Newton steps
// (2^-53); the compiler's own expansion under -freciprocal-math runs three.  0 and infinity give NaN, as
// there.
PDEV double frcp(const double x)
{
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}
// sqrt(x) and 1/sqrt(x) of a positive, normal x: v_rsq_f64 seed, one coupled Goldschmidt step, one
// residual correction of the root, one Newton step of the reciprocal root against it (both 2^-53 measured).  Against the library
// form this drops the range scaling (operands are squares and ratios of cgs-scale state variables, far
// from the subnormal and overflow ranges), the second correction and the zero/infinity fix-up:
// x = 0 gives NaN, so callers pass operands that are positive by construction.
PDEV void sqrt_rsqrt_pos(const double x, double &root, double &rroot)
{
  const double y = __builtin_amdgcn_rsq(x);
  double gq = x * y, hq = 0.5 * y;
  const double r = __builtin_fma(-hq, gq, 0.5);
  gq = __builtin_fma(gq, r, gq);
  hq = __builtin_fma(hq, r, hq);
  const double d = __builtin_fma(-gq, gq, x);
  root = __builtin_fma(d, hq, gq);
  const double r0 = hq + hq;   // 2^-48; one Newton step against the finished root
  rroot = __builtin_fma(r0, __builtin_fma(-root, r0, 1.0), r0);
}
// The same with one refinement step less -- 1/x to 2^-48.8, the roots to 2^-47 (about 30 ulp) -- for the MHD
// interface flux of the fast build only (HLLD wave speeds and star states, the FKJ98 viscosity coefficient): 14 of
// these per Riemann solve, 3.5 solves per cell and stage.  Measured on the long reference runs
// (profiles/tools/fast_margins.py): end-state L2 / refvec of the MHD blast 3e-15 -> 2e-14 against a gate of 1e-10;
// 512^3 GLM-MHD step 24.97 -> 24.27 ms.  Cell update, time-step reduction and every Euler path keep the full forms
// (the Euler shock runs amplify the difference to 2-5e-11 over 171-400 steps: too close to the gate).
PDEV double frcp_r(const double x)
{
  const double r = __builtin_amdgcn_rcp(x);
  return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
}
PDEV void sqrt_rsqrt_pos_r(const double x, double &root, double &rroot)
{
  const double y = __builtin_amdgcn_rsq(x);
  const double gq = x * y, hq = 0.5 * y;
  const double r = __builtin_fma(-hq, gq, 0.5);
  root = __builtin_fma(gq, r, gq);
  const double h1 = __builtin_fma(hq, r, hq);
  rroot = h1 + h1;
}
PDEV double sqrt_pos_r(const double x)
{
  double s, rs;
  sqrt_rsqrt_pos_r(x, s, rs);
  return s;
}
PDEV double sqrt_pos(const double x)
{
  double s, rs;
  sqrt_rsqrt_pos(x, s, rs);
  return s;
}
// root of a quantity that is positive for every valid state (density, gamma p / rho, ...); max / min
// of finite operands: the fast forms in the fast build, the reference's in the strict build
PDEV double psqrt(const double x) { return sqrt_pos(x); }
PDEV double prcp(const double x) { return frcp(x); }
PDEV double pmax(const double a, const double b) { return fmx(a, b); }
PDEV double pmin(const double a, const double b) { return fmn(a, b); }
#else
PDEV double psqrt(const double x) { return sqrt(x); }
PDEV double prcp(const double x) { return 1.0 / x; }   // (only used in fast-build branches)
PDEV double pmax(const double a, const double b) { return dmax(a, b); }
PDEV double pmin(const double a, const double b) { return dmin(a, b); }
#endif

template <int EQ>
struct EqBase {
  static constexpr int value = (EQ == EQEUL) ? 5 : ((EQ == EQMHD) ? 8 : 9);
};

// SoA variable index of sweep-frame slot s for a sweep along AXIS
template <int AXIS>
__host__ __device__ constexpr int gvar(int s)
{
  return (s >= 2 && s <= 4) ? 2 + ((s - 2) + AXIS) % 3 : ((s >= 5 && s <= 7) ? 5 + ((s - 5) + AXIS) % 3 : s);
}

// microphysics hooks (mp_only_cooling.cpp:81-85,255-280; microphysics_base.cpp:80-133)
struct MPd {
  int present;
  double Mu_tot_over_kB;
};

// pconst.equalD: constants.cpp:44-66
PDEV bool equalD(const double a, const double b)
{
  if (a == b) return true;
  if (fabs(a) + fabs(b) < PION_TINYVALUE) return true;
#ifdef PION_FAST_MATH
  if (fabs(a - b) < PION_SMALLVALUE * (fabs(a) + fabs(b) + PION_TINYVALUE)) return true;  // no division
#else
  if ((fabs(a - b) / (fabs(a) + fabs(b) + PION_TINYVALUE)) < PION_SMALLVALUE) return true;
#endif
  return false;
}

template <int EQ, int NTR>
struct Eqn {
  static constexpr int BASE = EqBase<EQ>::value;
  static constexpr int NV = BASE + NTR;
  static constexpr bool MHD = (EQ != EQEUL);

  // ---- Euler ------------------------------------------------------------
  static PDEV void euler_PtoU(const double *p, double *u, const double g)
  {
    u[uRHO] = p[qRO];
    u[uMN] = p[qRO] * p[qVN];
    u[uMT1] = p[qRO] * p[qVT1];
    u[uMT2] = p[qRO] * p[qVT2];
    u[uERG] = p[qRO] * (p[qVN] * p[qVN] + p[qVT1] * p[qVT1] + p[qVT2] * p[qVT2]) * 0.5 + p[qPG] / (g - 1.);
  }
#ifdef PION_FAST_MATH
  static PDEV double chydro(const double *p, const double g) { return sqrt_pos(g * p[qPG] * frcp(p[qRO])); }
#else
  static PDEV double chydro(const double *p, const double g) { return sqrt(g * p[qPG] / p[qRO]); }
#endif
  static PDEV void euler_PUtoFlux(const double *p, const double *u, double *f)
  {
    f[uRHO] = u[uMN];
    f[uMN] = u[uMN] * p[qVN] + p[qPG];
    f[uMT1] = u[uMN] * p[qVT1];
    f[uMT2] = u[uMN] * p[qVT2];
    f[uERG] = p[qVN] * (u[uERG] + p[qPG]);
  }
  static PDEV void euler_UtoFlux(const double *u, double *f, const double g)
  {
#ifdef PION_FAST_MATH
    {
      const double iro = frcp(u[uRHO]), vn = u[uMN] * iro;
      const double pgf = (g - 1.) * (u[uERG] - (u[uMN] * u[uMN] + u[uMT1] * u[uMT1] + u[uMT2] * u[uMT2]) * 0.5 * iro);
      f[uRHO] = u[uMN];
      f[uMN] = u[uMN] * vn + pgf;
      f[uMT1] = vn * u[uMT1];
      f[uMT2] = vn * u[uMT2];
      f[uERG] = vn * (u[uERG] + pgf);
      return;
    }
#endif
    double pg = (g - 1.) * (u[uERG] - (u[uMN] * u[uMN] + u[uMT1] * u[uMT1] + u[uMT2] * u[uMT2]) * 0.5 / u[uRHO]);
    f[uRHO] = u[uMN];
    f[uMN] = u[uMN] * u[uMN] / u[uRHO] + pg;
    f[uMT1] = u[uMN] * u[uMT1] / u[uRHO];
    f[uMT2] = u[uMN] * u[uMT2] / u[uRHO];
    f[uERG] = u[uMN] * (u[uERG] + pg) / u[uRHO];
  }
  static PDEV double Enthalpy(const double *p, const double g)
  {
    return (0.5 * (p[qVN] * p[qVN] + p[qVT1] * p[qVT1] + p[qVT2] * p[qVT2]) + g * p[qPG] / (g - 1.0) / p[qRO]);
  }

  // ---- ideal MHD --------------------------------------------------------
  static PDEV void mhd_PtoU(const double *p, double *u, const double g)
  {
    u[uRHO] = p[qRO];
    u[uMN] = p[qRO] * p[qVN];
    u[uMT1] = p[qRO] * p[qVT1];
    u[uMT2] = p[qRO] * p[qVT2];
    u[uBN] = p[qBN];
    u[uBT1] = p[qBT1];
    u[uBT2] = p[qBT2];
    u[uERG] = (p[qRO] * (p[qVN] * p[qVN] + p[qVT1] * p[qVT1] + p[qVT2] * p[qVT2]) * 0.5) + (p[qPG] / (g - 1.)) +
              ((u[uBN] * u[uBN] + u[uBT1] * u[uBT1] + u[uBT2] * u[uBT2]) * 0.5);
  }
  static PDEV double cfast(const double *p, const double g)
  {
    double ch = chydro(p, g);
    double temp1 = ch * ch + (p[qBN] * p[qBN] + p[qBT1] * p[qBT1] + p[qBT2] * p[qBT2]) / p[qRO];
    double temp2 = 4. * ch * ch * p[qBN] * p[qBN] / p[qRO];
    temp2 = dmax(PION_MACHINEACCURACY, temp1 * temp1 - temp2);
    return (sqrt((temp1 + sqrt(temp2)) / 2.));
  }
  static PDEV double cfast_components(const double cfRO, const double cfPG, const double cfBX, const double cfBY,
                                      const double cfBZ, const double g)
  {
#ifdef PION_FAST_MATH
    // fast build: a^2 = gamma p / rho directly (the strict form squares its square root), one reciprocal
    const double ir = frcp(cfRO);
    const double a2 = g * cfPG * ir;
    const double temp1 = a2 + (cfBX * cfBX + cfBY * cfBY + cfBZ * cfBZ) * ir;
    const double temp2 = fmx(PION_MACHINEACCURACY, temp1 * temp1 - 4. * a2 * cfBX * cfBX * ir);
    return sqrt_pos((temp1 + sqrt_pos(temp2)) * 0.5);
#else
    double ch = sqrt(g * cfPG / cfRO);
    double temp1 = ch * ch + (cfBX * cfBX + cfBY * cfBY + cfBZ * cfBZ) / cfRO;
    double temp2 = 4. * ch * ch * cfBX * cfBX / cfRO;
    temp2 = dmax(PION_MACHINEACCURACY, temp1 * temp1 - temp2);
    return (sqrt((temp1 + sqrt(temp2)) / 2.));
#endif
  }
  static PDEV void mhd_PUtoFlux(const double *p, const double *u, double *f)
  {
    double pm = (u[uBN] * u[uBN] + u[uBT1] * u[uBT1] + u[uBT2] * u[uBT2]) / 2.;
    f[uRHO] = u[uMN];
    f[uMN] = u[uMN] * p[qVN] + p[qPG] + pm - u[uBN] * u[uBN];
    f[uMT1] = u[uMN] * p[qVT1] - u[uBN] * u[uBT1];
    f[uMT2] = u[uMN] * p[qVT2] - u[uBN] * u[uBT2];
    f[uERG] = p[qVN] * (u[uERG] + p[qPG] + pm) - u[uBN] * (p[qVN] * u[uBN] + p[qVT1] * u[uBT1] + p[qVT2] * u[uBT2]);
    f[uBN] = 0.;
    f[uBT1] = p[qVN] * p[qBT1] - p[qVT1] * p[qBN];
    f[uBT2] = p[qVN] * p[qBT2] - p[qVT2] * p[qBN];
  }
  static PDEV void mhd_UtoFlux(const double *u, double *f, const double g)
  {
    double pm = (u[uBN] * u[uBN] + u[uBT1] * u[uBT1] + u[uBT2] * u[uBT2]) / 2.;
    double pg = (g - 1.) * (u[uERG] - (u[uMN] * u[uMN] + u[uMT1] * u[uMT1] + u[uMT2] * u[uMT2]) / (2. * u[uRHO]) - pm);
    f[uRHO] = u[uMN];
    f[uMN] = u[uMN] * u[uMN] / u[uRHO] + pg + pm - u[uBN] * u[uBN];
    f[uMT1] = u[uMN] * u[uMT1] / u[uRHO] - u[uBN] * u[uBT1];
    f[uMT2] = u[uMN] * u[uMT2] / u[uRHO] - u[uBN] * u[uBT2];
    f[uERG] = u[uMN] * (u[uERG] + pg + pm) / u[uRHO] -
              u[uBN] * (u[uMN] * u[uBN] + u[uMT1] * u[uBT1] + u[uMT2] * u[uBT2]) / u[uRHO];
    f[uBN] = 0.;
    f[uBT1] = (u[uMN] * u[uBT1] - u[uMT1] * u[uBN]) / u[uRHO];
    f[uBT2] = (u[uMN] * u[uBT2] - u[uMT2] * u[uBN]) / u[uRHO];
  }
  static PDEV double mhd_Ptot(const double *p)
  {
    return (p[qPG] + 0.5 * (p[qBN] * p[qBN] + p[qBT1] * p[qBT1] + p[qBT2] * p[qBT2]));
  }

  // ---- "virtual" versions: what FV_solver_* dispatches to -----------------
  static PDEV void PtoU(const double *p, double *u, const double g)
  {
    if constexpr (EQ == EQEUL) {
      euler_PtoU(p, u, g);
    }
    else if constexpr (EQ == EQMHD) {
      mhd_PtoU(p, u, g);
    }
    else {
      u[uPSI] = p[qSI];
      mhd_PtoU(p, u, g);
      u[uERG] += 0.5 * u[uPSI] * u[uPSI];
    }
#pragma unroll
    for (int t = 0; t < NTR; t++) u[BASE + t] = p[BASE + t] * p[qRO];
  }
  // pressure repairs: eqns_hydro_adiabatic.cpp:127-198 / eqns_mhd_adiabatic.cpp:139-225
  static PDEV void check_pressure(double *p, const double MinTemp, const MPd &mp, int &err)
  {
    if (p[qRO] <= 0.0) err |= ERR_NEG_DENSITY;  // reference: rep.error -> exit(1)
    if (p[qPG] <= 0.0) {
      if (mp.present) p[qPG] = p[qRO] * MinTemp / mp.Mu_tot_over_kB;
      else p[qPG] = 0.01 * p[qRO];
    }
    else if (mp.present && ((p[qPG] * mp.Mu_tot_over_kB / p[qRO]) < MinTemp)) {
      p[qPG] = p[qRO] * MinTemp / mp.Mu_tot_over_kB;
    }
  }
  static PDEV void UtoP(const double *u, double *p, const double MinTemp, const double g, const MPd &mp, int &err)
  {
#pragma unroll
    for (int t = 0; t < NTR; t++) p[BASE + t] = u[BASE + t] / u[uRHO];
    if constexpr (EQ == EQGLM) p[qSI] = u[uPSI];
    p[qRO] = u[uRHO];
#ifdef PION_FAST_MATH
    const double iro = frcp(u[uRHO]);
    p[qVN] = u[uMN] * iro;
    p[qVT1] = u[uMT1] * iro;
    p[qVT2] = u[uMT2] * iro;
#else
    p[qVN] = u[uMN] / u[uRHO];
    p[qVT1] = u[uMT1] / u[uRHO];
    p[qVT2] = u[uMT2] / u[uRHO];
#endif
    if constexpr (EQ == EQEUL) {
      p[qPG] = (g - 1.0) * (u[uERG] - p[qRO] * (p[qVN] * p[qVN] + p[qVT1] * p[qVT1] + p[qVT2] * p[qVT2]) / 2.0);
    }
    else if constexpr (EQ == EQMHD) {
      p[qPG] = (g - 1) * (u[uERG] - p[qRO] * (p[qVN] * p[qVN] + p[qVT1] * p[qVT1] + p[qVT2] * p[qVT2]) / 2. -
                          (u[uBN] * u[uBN] + u[uBT1] * u[uBT1] + u[uBT2] * u[uBT2]) / 2.);
    }
    else {
      p[qPG] = (g - 1.0) * (u[uERG] - p[qRO] * (p[qVN] * p[qVN] + p[qVT1] * p[qVT1] + p[qVT2] * p[qVT2]) * 0.5 -
                            0.5 * u[uPSI] * u[uPSI] - (u[uBN] * u[uBN] + u[uBT1] * u[uBT1] + u[uBT2] * u[uBT2]) * 0.5);
    }
    if constexpr (MHD) {
      p[qBN] = u[uBN];
      p[qBT1] = u[uBT1];
      p[qBT2] = u[uBT2];
    }
    check_pressure(p, MinTemp, mp, err);
  }
  // eqns_base::PtoFlux (virtual PtoU+PUtoFlux) for Euler, eqns_mhd_ideal::PtoFlux for MHD.
  // Tracer entries are left untouched (they are overwritten by the interface tracer flux).
  static PDEV void PtoFlux(const double *p, double *f, const double g)
  {
    double u[NV];
    if constexpr (EQ == EQEUL) {
      euler_PtoU(p, u, g);
      euler_PUtoFlux(p, u, f);
    }
    else {
      mhd_PtoU(p, u, g);
      mhd_PUtoFlux(p, u, f);
    }
  }
  static PDEV void UtoFlux(const double *u, double *f, const double g)
  {
    if constexpr (EQ == EQEUL) euler_UtoFlux(u, f, g);
    else mhd_UtoFlux(u, f, g);
#pragma unroll
    for (int t = 0; t < NTR; t++) f[BASE + t] = u[BASE + t] * f[uRHO] / u[uRHO];
  }
  static PDEV double maxspeed(const double *p, const double g) { return MHD ? cfast(p, g) : chydro(p, g); }

  // microphysics_base::sCMA (microphysics_base.cpp:80-133), tracers only
  static PDEV void apply_sCMA(double *p)
  {
#pragma unroll
    for (int t = 0; t < NTR; t++) {
      const double x = p[BASE + t];
      const double corr = (x > 1.0) ? 1.0 / x : 1.0;
      p[BASE + t] = x * corr;
    }
  }
};

}  // namespace pion
#endif
