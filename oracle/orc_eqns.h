// orc_eqns.h -- ORACLE (test infrastructure, never shipped, never on the product path).
//
// CPU restatement of PION's equations classes: primitive<->conserved<->flux
// conversions, wave speeds and pressure repairs for Euler, ideal MHD and
// GLM-MHD, with the reference's direction handling by index permutation
// (source/equations/eqns_base.cpp:94-132).  Written from the reference's
// behaviour, operation order kept so that results are bit-identical to the
// reference when compiled with -ffp-contract=off.
//
// Follows: source/equations/eqns_hydro_adiabatic.cpp:89-453,
//          source/equations/eqns_mhd_adiabatic.cpp:79-660,
//          source/spatial_solvers/solver_eqn_hydro_adi.cpp:211-273 (tracers),
//          source/spatial_solvers/solver_eqn_mhd_adi.cpp:288-366, 846-904.
#ifndef ORC_EQNS_H
#define ORC_EQNS_H

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

namespace orc {

// constants.h:150-157, 336-339
constexpr double SMALLVALUE = 1.0e-12;
constexpr double MACHINEACCURACY = 5.e-16;
constexpr double TINYVALUE = 1.0e-100;
constexpr double VERY_TINY_VALUE = 1.0e-200;
constexpr double BASEPG = 1.e-5;
constexpr double BASE_RHO = 1.0e-5;

enum { RO = 0, PG = 1, VX = 2, VY = 3, VZ = 4, BX = 5, BY = 6, BZ = 7, SI = 8 };
enum { RHO = 0, ERG = 1, MMX = 2, MMY = 3, MMZ = 4, BBX = 5, BBY = 6, BBZ = 7, PSI = 8 };
enum { EQEUL = 1, EQMHD = 2, EQGLM = 3 };
constexpr int MAXNV = 16;

struct physics_error : public std::runtime_error {
  explicit physics_error(const std::string &s) : std::runtime_error(s) {}
};

// constants.cpp:44-66
inline bool equalD(const double a, const double b)
{
  if (a == b) return true;
  if (std::fabs(a) + std::fabs(b) < TINYVALUE) return true;
  if ((std::fabs(a - b) / (std::fabs(a) + std::fabs(b) + TINYVALUE)) < SMALLVALUE) return true;
  return false;
}

// Microphysics hooks needed by UtoP / sCMA (microphysics_base.h:53-337; only the
// mp_only_cooling behaviour is restated, mp_only_cooling.cpp:81-85,255-280).
struct MPHooks {
  bool present = false;
  double Mu_tot_over_kB = 0.0;
  int nvar = 0, ntracer = 0;
  double Temperature(const double *p) const { return p[PG] * Mu_tot_over_kB / p[RO]; }
  void Set_Temp(double *p, const double T) const { p[PG] = p[RO] * T / Mu_tot_over_kB; }
  // microphysics_base.cpp:80-133.  No "X_" element tracers are supported, so the
  // element renormalisation is a no-op; the (p<0)->0 assignment is overwritten
  // by the next line in the reference and is reproduced as such.
  void sCMA(double *corrector, const double *p_in) const
  {
    for (int i = 0; i < nvar; i++) corrector[i] = 1.0;
    for (int v = 0; v < ntracer; v++) {
      const int i = nvar - ntracer + v;
      corrector[i] = (p_in[i] < 0.0) ? 0.0 : 1.0;
      corrector[i] = (p_in[i] > 1.0) ? 1.0 / p_in[i] : 1.0;
    }
  }
};

// One equations object = eqns_base + eqns_Euler / eqns_mhd_ideal / eqns_mhd_mixedGLM
// + the tracer extensions of the FV_solver_* classes.
struct Eqns {
  int eqntype = EQEUL;
  int nvar = 5;
  int ntr = 0;
  int eqTR[MAXNV];
  double gamma = 5. / 3.;
  double refvec[MAXNV];  // eq_refvec after SetAvgState
  double GLM_chyp = 0.0, GLM_cr = 0.0;
  const MPHooks *MP = nullptr;
  // direction-dependent indices (eqns_base.cpp:94-132)
  int dir = 0;
  int eqVX = VX, eqVY = VY, eqVZ = VZ, eqBX = BX, eqBY = BY, eqBZ = BZ;
  // conserved indices equal the primitive ones numerically (constants.h:256-281)
  int eqMMX = MMX, eqMMY = MMY, eqMMZ = MMZ, eqBBX = BBX, eqBBY = BBY, eqBBZ = BBZ;
  static constexpr int eqRO = RO, eqPG = PG, eqRHO = RHO, eqERG = ERG, eqSI = SI, eqPSI = PSI;

  bool is_mhd() const { return eqntype == EQMHD || eqntype == EQGLM; }

  void init(int eqt, int nv, int ntracer, double g)
  {
    eqntype = eqt;
    nvar = nv;
    ntr = ntracer;
    gamma = g;
    for (int i = 0; i < ntr; i++) eqTR[i] = nvar - ntr + i;  // solver_eqn_base.cpp:76-80
    for (int v = 0; v < MAXNV; v++) refvec[v] = 0.0;
    SetDirection(0);
  }

  void SetDirection(const int d)
  {
    dir = d;
    switch (d) {
      case 0:
        eqVX = VX; eqVY = VY; eqVZ = VZ; eqBX = BX; eqBY = BY; eqBZ = BZ;
        break;
      case 1:
        eqVX = VY; eqVY = VZ; eqVZ = VX; eqBX = BY; eqBY = BZ; eqBZ = BX;
        break;
      case 2:
        eqVX = VZ; eqVY = VX; eqVZ = VY; eqBX = BZ; eqBY = BX; eqBZ = BY;
        break;
      default:
        throw physics_error("bad direction in SetDirection");
    }
    eqMMX = eqVX; eqMMY = eqVY; eqMMZ = eqVZ;
    eqBBX = eqBX; eqBBY = eqBY; eqBBZ = eqBZ;
  }

  // ---- Euler (eqns_hydro_adiabatic.cpp) ---------------------------------
  void euler_PtoU(const double *p, double *u, const double g) const
  {
    u[eqRHO] = p[eqRO];
    u[eqMMX] = p[eqRO] * p[eqVX];
    u[eqMMY] = p[eqRO] * p[eqVY];
    u[eqMMZ] = p[eqRO] * p[eqVZ];
    u[eqERG] = p[eqRO] * (p[eqVX] * p[eqVX] + p[eqVY] * p[eqVY] + p[eqVZ] * p[eqVZ]) * 0.5 +
               p[eqPG] / (g - 1.);
  }
  int euler_UtoP(const double *u, double *p, const double MinTemp, const double g) const
  {
    p[eqRO] = u[eqRHO];
    p[eqVX] = u[eqMMX] / u[eqRHO];
    p[eqVY] = u[eqMMY] / u[eqRHO];
    p[eqVZ] = u[eqMMZ] / u[eqRHO];
    p[eqPG] = (g - 1.0) *
              (u[eqERG] -
               p[eqRO] * (p[eqVX] * p[eqVX] + p[eqVY] * p[eqVY] + p[eqVZ] * p[eqVZ]) / 2.0);
    if (p[eqRO] <= 0.0) {
      // eqns_hydro_adiabatic.cpp:140-147: rep.error -> exit(1)
      throw physics_error("Negative density (eqns_Euler::UtoP)");
    }
    // SET_NEGATIVE_PRESSURE_TO_FIXED_TEMPERATURE (functionality_flags.h)
    if (p[eqPG] <= 0.0) {
      if (MP && MP->present) MP->Set_Temp(p, MinTemp);
      else p[eqPG] = 0.01 * p[eqRO];
    }
    else if (MP && MP->present && (MP->Temperature(p) < MinTemp)) {
      MP->Set_Temp(p, MinTemp);
    }
    return 0;
  }
  double chydro(const double *p, const double g) const { return std::sqrt(g * p[eqPG] / p[eqRO]); }
  void euler_PUtoFlux(const double *p, const double *u, double *f) const
  {
    f[eqRHO] = u[eqMMX];
    f[eqMMX] = u[eqMMX] * p[eqVX] + p[eqPG];
    f[eqMMY] = u[eqMMX] * p[eqVY];
    f[eqMMZ] = u[eqMMX] * p[eqVZ];
    f[eqERG] = p[eqVX] * (u[eqERG] + p[eqPG]);
  }
  void euler_UtoFlux(const double *u, double *f, const double g) const
  {
    double pg = (g - 1.) *
                (u[eqERG] - (u[eqMMX] * u[eqMMX] + u[eqMMY] * u[eqMMY] + u[eqMMZ] * u[eqMMZ]) *
                                0.5 / u[eqRHO]);
    f[eqRHO] = u[eqMMX];
    f[eqMMX] = u[eqMMX] * u[eqMMX] / u[eqRHO] + pg;
    f[eqMMY] = u[eqMMX] * u[eqMMY] / u[eqRHO];
    f[eqMMZ] = u[eqMMX] * u[eqMMZ] / u[eqRHO];
    f[eqERG] = u[eqMMX] * (u[eqERG] + pg) / u[eqRHO];
  }
  double Enthalpy(const double *p, const double g) const
  {
    return (0.5 * (p[eqVX] * p[eqVX] + p[eqVY] * p[eqVY] + p[eqVZ] * p[eqVZ]) +
            g * p[eqPG] / (g - 1.0) / p[eqRO]);
  }

  // ---- ideal MHD (eqns_mhd_adiabatic.cpp) -------------------------------
  void mhd_PtoU(const double *p, double *u, const double g) const
  {
    u[eqRHO] = p[eqRO];
    u[eqMMX] = p[eqRO] * p[eqVX];
    u[eqMMY] = p[eqRO] * p[eqVY];
    u[eqMMZ] = p[eqRO] * p[eqVZ];
    u[eqBBX] = p[eqBX];
    u[eqBBY] = p[eqBY];
    u[eqBBZ] = p[eqBZ];
    u[eqERG] = (p[eqRO] * (p[eqVX] * p[eqVX] + p[eqVY] * p[eqVY] + p[eqVZ] * p[eqVZ]) * 0.5) +
               (p[eqPG] / (g - 1.)) +
               ((u[eqBBX] * u[eqBBX] + u[eqBBY] * u[eqBBY] + u[eqBBZ] * u[eqBBZ]) * 0.5);
  }
  int check_pressure(const double * /*u*/, double *p, const double MinTemp) const
  {
    if (p[eqRO] <= 0.0) {
      // eqns_mhd_adiabatic.cpp:154-158: rep.error -> exit(1)
      throw physics_error("Negative Density! Bugging out (eqns_mhd_ideal::check_pressure)");
    }
    if (p[eqPG] <= 0.0) {
      if (MP && MP->present) MP->Set_Temp(p, MinTemp);
      else p[eqPG] = 0.01 * p[eqRO];
    }
    else if (MP && MP->present && (MP->Temperature(p) < MinTemp)) {
      MP->Set_Temp(p, MinTemp);
    }
    return 0;
  }
  int mhd_UtoP(const double *u, double *p, const double MinTemp, const double g) const
  {
    p[eqRO] = u[eqRHO];
    p[eqVX] = u[eqMMX] / u[eqRHO];
    p[eqVY] = u[eqMMY] / u[eqRHO];
    p[eqVZ] = u[eqMMZ] / u[eqRHO];
    p[eqPG] = (g - 1) *
              (u[eqERG] -
               p[eqRO] * (p[eqVX] * p[eqVX] + p[eqVY] * p[eqVY] + p[eqVZ] * p[eqVZ]) / 2. -
               (u[eqBBX] * u[eqBBX] + u[eqBBY] * u[eqBBY] + u[eqBBZ] * u[eqBBZ]) / 2.);
    p[eqBX] = u[eqBBX];
    p[eqBY] = u[eqBBY];
    p[eqBZ] = u[eqBBZ];
    return check_pressure(u, p, MinTemp);
  }
  double cfast(const double *p, const double g) const
  {
    double ch = chydro(p, g);
    double temp1 = ch * ch + (p[eqBX] * p[eqBX] + p[eqBY] * p[eqBY] + p[eqBZ] * p[eqBZ]) / p[eqRO];
    double temp2 = 4. * ch * ch * p[eqBX] * p[eqBX] / p[eqRO];
    temp2 = std::max(MACHINEACCURACY, temp1 * temp1 - temp2);
    return (std::sqrt((temp1 + std::sqrt(temp2)) / 2.));
  }
  static double cfast_components(const double cfRO, const double cfPG, const double cfBX,
                                 const double cfBY, const double cfBZ, const double g)
  {
    double ch = std::sqrt(g * cfPG / cfRO);
    double temp1 = ch * ch + (cfBX * cfBX + cfBY * cfBY + cfBZ * cfBZ) / cfRO;
    double temp2 = 4. * ch * ch * cfBX * cfBX / cfRO;
    temp2 = std::max(MACHINEACCURACY, temp1 * temp1 - temp2);
    return (std::sqrt((temp1 + std::sqrt(temp2)) / 2.));
  }
  void mhd_PUtoFlux(const double *p, const double *u, double *f) const
  {
    double pm = (u[eqBBX] * u[eqBBX] + u[eqBBY] * u[eqBBY] + u[eqBBZ] * u[eqBBZ]) / 2.;
    f[eqRHO] = u[eqMMX];
    f[eqMMX] = u[eqMMX] * p[eqVX] + p[eqPG] + pm - u[eqBBX] * u[eqBBX];
    f[eqMMY] = u[eqMMX] * p[eqVY] - u[eqBBX] * u[eqBBY];
    f[eqMMZ] = u[eqMMX] * p[eqVZ] - u[eqBBX] * u[eqBBZ];
    f[eqERG] = p[eqVX] * (u[eqERG] + p[eqPG] + pm) -
               u[eqBBX] * (p[eqVX] * u[eqBBX] + p[eqVY] * u[eqBBY] + p[eqVZ] * u[eqBBZ]);
    f[eqBBX] = 0.;
    f[eqBBY] = p[eqVX] * p[eqBY] - p[eqVY] * p[eqBX];
    f[eqBBZ] = p[eqVX] * p[eqBZ] - p[eqVZ] * p[eqBX];
  }
  void mhd_UtoFlux(const double *u, double *f, const double g) const
  {
    double pm = (u[eqBBX] * u[eqBBX] + u[eqBBY] * u[eqBBY] + u[eqBBZ] * u[eqBBZ]) / 2.;
    double pg = (g - 1.) * (u[eqERG] -
                            (u[eqMMX] * u[eqMMX] + u[eqMMY] * u[eqMMY] + u[eqMMZ] * u[eqMMZ]) /
                                (2. * u[eqRHO]) -
                            pm);
    f[eqRHO] = u[eqMMX];
    f[eqMMX] = u[eqMMX] * u[eqMMX] / u[eqRHO] + pg + pm - u[eqBBX] * u[eqBBX];
    f[eqMMY] = u[eqMMX] * u[eqMMY] / u[eqRHO] - u[eqBBX] * u[eqBBY];
    f[eqMMZ] = u[eqMMX] * u[eqMMZ] / u[eqRHO] - u[eqBBX] * u[eqBBZ];
    f[eqERG] = u[eqMMX] * (u[eqERG] + pg + pm) / u[eqRHO] -
               u[eqBBX] * (u[eqMMX] * u[eqBBX] + u[eqMMY] * u[eqBBY] + u[eqMMZ] * u[eqBBZ]) /
                   u[eqRHO];
    f[eqBBX] = 0.;
    f[eqBBY] = (u[eqMMX] * u[eqBBY] - u[eqMMY] * u[eqBBX]) / u[eqRHO];
    f[eqBBZ] = (u[eqMMX] * u[eqBBZ] - u[eqMMZ] * u[eqBBX]) / u[eqRHO];
  }
  double mhd_Ptot(const double *p) const
  {
    return (p[eqPG] + 0.5 * (p[eqBX] * p[eqBX] + p[eqBY] * p[eqBY] + p[eqBZ] * p[eqBZ]));
  }
  // eqns_mhd_ideal::rotate (eqns_mhd_adiabatic.cpp:372-410), XX -> newdir only
  void mhd_rotate_from_X(double *vec, const int finaldir) const
  {
    if (finaldir == 0) return;
    double v[MAXNV];
    for (int i = 0; i < nvar; i++) v[i] = vec[i];
    int offset = (finaldir - 0 + 3) % 3;
    if (offset == 1) {
      v[eqVX] = vec[eqVY]; v[eqVY] = vec[eqVZ]; v[eqVZ] = vec[eqVX];
      v[eqBX] = vec[eqBY]; v[eqBY] = vec[eqBZ]; v[eqBZ] = vec[eqBX];
    }
    else {
      v[eqVX] = vec[eqVZ]; v[eqVY] = vec[eqVX]; v[eqVZ] = vec[eqVY];
      v[eqBX] = vec[eqBZ]; v[eqBY] = vec[eqBX]; v[eqBZ] = vec[eqBY];
    }
    for (int i = 0; i < nvar; i++) vec[i] = v[i];
  }

  // ---- virtual dispatch of the FV_solver classes -------------------------
  // FV_solver_Hydro_Euler::PtoU etc. (solver_eqn_hydro_adi.cpp:211-273),
  // FV_solver_mhd_ideal_adi (solver_eqn_mhd_adi.cpp:288-366),
  // FV_solver_mhd_mixedGLM_adi (solver_eqn_mhd_adi.cpp:846-904) with
  // eqns_mhd_mixedGLM (eqns_mhd_adiabatic.cpp:598-660).
  void PtoU(const double *p, double *u, const double g) const
  {
    if (eqntype == EQEUL) {
      for (int t = 0; t < ntr; t++) u[eqTR[t]] = p[eqTR[t]] * p[eqRO];
      euler_PtoU(p, u, g);
    }
    else if (eqntype == EQMHD) {
      mhd_PtoU(p, u, g);
      for (int t = 0; t < ntr; t++) u[eqTR[t]] = p[eqTR[t]] * p[eqRO];
    }
    else {
      u[eqPSI] = p[eqSI];
      mhd_PtoU(p, u, g);
      u[eqERG] += 0.5 * u[eqPSI] * u[eqPSI];
      for (int t = 0; t < ntr; t++) u[eqTR[t]] = p[eqTR[t]] * p[eqRO];
    }
  }
  int UtoP(const double *u, double *p, const double MinTemp, const double g) const
  {
    for (int t = 0; t < ntr; t++) p[eqTR[t]] = u[eqTR[t]] / u[eqRHO];
    if (eqntype == EQEUL) return euler_UtoP(u, p, MinTemp, g);
    if (eqntype == EQMHD) return mhd_UtoP(u, p, MinTemp, g);
    // eqns_mhd_mixedGLM::UtoP
    p[eqSI] = u[eqPSI];
    p[eqRO] = u[eqRHO];
    p[eqVX] = u[eqMMX] / u[eqRHO];
    p[eqVY] = u[eqMMY] / u[eqRHO];
    p[eqVZ] = u[eqMMZ] / u[eqRHO];
    p[eqPG] = (g - 1.0) *
              (u[eqERG] -
               p[eqRO] * (p[eqVX] * p[eqVX] + p[eqVY] * p[eqVY] + p[eqVZ] * p[eqVZ]) * 0.5 -
               0.5 * u[eqPSI] * u[eqPSI] -
               (u[eqBBX] * u[eqBBX] + u[eqBBY] * u[eqBBY] + u[eqBBZ] * u[eqBBZ]) * 0.5);
    p[eqBX] = u[eqBBX];
    p[eqBY] = u[eqBBY];
    p[eqBZ] = u[eqBBZ];
    return check_pressure(u, p, MinTemp);
  }
  // f must hold the values the caller's array had before the call: the Euler
  // variant reads f[eqRHO] for the tracers BEFORE setting it
  // (solver_eqn_hydro_adi.cpp:243-251).
  void PUtoFlux(const double *p, const double *u, double *f) const
  {
    if (eqntype == EQEUL) {
      for (int t = 0; t < ntr; t++) f[eqTR[t]] = p[eqTR[t]] * f[eqRHO];
      euler_PUtoFlux(p, u, f);
    }
    else {
      mhd_PUtoFlux(p, u, f);
      for (int t = 0; t < ntr; t++) f[eqTR[t]] = p[eqTR[t]] * f[eqRHO];
    }
  }
  void UtoFlux(const double *u, double *f, const double g) const
  {
    if (eqntype == EQEUL) euler_UtoFlux(u, f, g);
    else mhd_UtoFlux(u, f, g);
    for (int t = 0; t < ntr; t++) f[eqTR[t]] = u[eqTR[t]] * f[eqRHO] / u[eqRHO];
  }
  // eqns_base::PtoFlux (eqns_base.cpp:236-246) / eqns_mhd_ideal::PtoFlux (:357-366)
  void PtoFlux(const double *p, double *f, const double g) const
  {
    double u[MAXNV];
    if (eqntype == EQEUL) {
      PtoU(p, u, g);
      PUtoFlux(p, u, f);
    }
    else {
      mhd_PtoU(p, u, g);
      mhd_PUtoFlux(p, u, f);
    }
  }
  double maxspeed(const double *p, const double g) const
  {
    return is_mhd() ? cfast(p, g) : chydro(p, g);
  }

  // SetAvgState: eqns_hydro_adiabatic.cpp:437-453, eqns_mhd_adiabatic.cpp:487-544
  void SetAvgState(const double *ms, const double g)
  {
    const int savedir = dir;
    SetDirection(0);
    if (eqntype == EQEUL) {
      for (int v = 0; v < nvar; v++) refvec[v] = ms[v];
      double refvel = chydro(refvec, g);
      refvec[eqVX] = refvec[eqVY] = refvec[eqVZ] = 0.1 * refvel;
    }
    else {
      for (int v = 0; v < 8; v++) refvec[v] = ms[v];
      double angle = refvec[eqBY] * refvec[eqBY] + refvec[eqBX] * refvec[eqBX];
      double refvel = 0.0;
      if (angle > 10. * MACHINEACCURACY) {
        angle = M_PI / 2. - std::asin(refvec[eqBY] / std::sqrt(angle));
        if (refvec[eqBX] < 0) angle = -angle;
        rotateXY(refvec, angle);
        refvel = cfast(refvec, gamma);
        rotateXY(refvec, -angle);
      }
      else refvel = cfast(refvec, gamma);
      double refB = std::sqrt(refvec[eqBX] * refvec[eqBX] + refvec[eqBY] * refvec[eqBY] +
                              refvec[eqBZ] * refvec[eqBZ]);
      refvec[eqVX] = refvec[eqVY] = refvec[eqVZ] = 0.1 * refvel;
      refvec[eqBX] = refvec[eqBY] = refvec[eqBZ] = refB;
    }
    SetDirection(savedir);
  }
  void rotateXY(double *v, double theta) const
  {
    double ct = std::cos(theta), st = std::sin(theta);
    double vx = v[eqVX] * ct - v[eqVY] * st;
    double vy = v[eqVX] * st + v[eqVY] * ct;
    v[eqVX] = vx;
    v[eqVY] = vy;
    vx = v[eqBX] * ct - v[eqBY] * st;
    vy = v[eqBX] * st + v[eqBY] * ct;
    v[eqBX] = vx;
    v[eqBY] = vy;
  }
};

}  // namespace orc
#endif
