// orc_riemann.h -- ORACLE (test infrastructure, never shipped, never on the product path).
//
// CPU restatement of PION's flux functions and of FV_solver_base::InterCellFlux.
// Each function cites the reference file:line it follows; the operation order
// is the reference's so that results are bit-identical with -ffp-contract=off.
#ifndef ORC_RIEMANN_H
#define ORC_RIEMANN_H

#include "orc_eqns.h"

namespace orc {

enum {
  FLUX_LF = 0, FLUX_RSlinear = 1, FLUX_RSexact = 2, FLUX_RShybrid = 3, FLUX_RSroe = 4,
  FLUX_RSroe_pv = 5, FLUX_FVS = 6, FLUX_RS_HLLD = 7, FLUX_RS_HLL = 8
};
enum { AV_NONE = 0, AV_FKJ98_1D = 1, AV_HCORRECTION = 3, AV_HCORR_FKJ98 = 4 };
enum { XN = 0, XP = 1 };

struct Solver : public Eqns {
  int gndim = 1;          // FV_gndim
  double cfl = 0.3;       // FV_cfl
  double FV_dt = 0.0;     // FV_dt (Setdt)
  double etav = 0.0;      // FV_etav == FV_etaB (solver_eqn_base.cpp:82)
  double HC_etamax = 0.0; // set by pre_calc_viscous_terms
  double MinTemperature = 0.0;
  // persistent scratch of the reference's solver objects whose stale contents
  // can leak into (unused) outputs: HLL_hydro::HD_FL/HD_FR
  double HD_FL[MAXNV], HD_FR[MAXNV];
  // riemann_Euler state that persists between calls (riemann.h)
  double rs_left[5], rs_right[5], rs_meanp[5], rs_pstar[5], cl = 0, cr = 0;

  Solver()
  {
    for (int v = 0; v < MAXNV; v++) HD_FL[v] = HD_FR[v] = 0.0;
    for (int v = 0; v < 5; v++) rs_left[v] = rs_right[v] = rs_meanp[v] = rs_pstar[v] = 0.0;
  }

  // ---------------------------------------------------------------------
  // Lax-Friedrichs: solver_eqn_base.cpp:109-141
  int get_LaxFriedrichs_flux(const double *l, const double *r, double *f, const double dx)
  {
    double u1[MAXNV], u2[MAXNV], f1[MAXNV], f2[MAXNV];
    for (int v = 0; v < MAXNV; v++) f1[v] = f2[v] = 0.0;
    PtoU(l, u1, gamma);
    PtoU(r, u2, gamma);
    UtoFlux(u1, f1, gamma);
    UtoFlux(u2, f2, gamma);
    for (int v = 0; v < nvar; v++)
      f[v] = 0.5 * (f1[v] + f2[v] + dx / FV_dt * (u1[v] - u2[v]) / gndim);
    if (ntr > 0) {
      if (f[eqRHO] >= 0.) {
        for (int t = 0; t < ntr; t++) f[eqTR[t]] = l[eqTR[t]] * f[eqRHO];
      }
      else {
        for (int t = 0; t < ntr; t++) f[eqTR[t]] = r[eqTR[t]] * f[eqRHO];
      }
    }
    return 0;
  }

  // ---------------------------------------------------------------------
  // Roe conserved-variable solver, symmetric version:
  // Roe_Hydro_ConservedVar_solver.cpp:129-247 and its pieces :303-597
  int Roe_flux_solver_symmetric(const double *left, const double *right, const double g,
                                const double hc_eta, double *out_pstar, double *out_flux)
  {
    const int eqHH = eqPG;
    double meanp[5], ul[5], ur[5], eval[5], strength[5], udiff[5], evec[5][5];
    // set_Roe_mean_state :303-342
    double rl = std::sqrt(left[eqRO]), rr = std::sqrt(right[eqRO]), lH = Enthalpy(left, g),
           rH = Enthalpy(right, g), denom = 1.0 / (rl + rr);
    meanp[eqRO] = rl * rr;
    meanp[eqVX] = (rl * left[eqVX] + rr * right[eqVX]) * denom;
    meanp[eqVY] = (rl * left[eqVY] + rr * right[eqVY]) * denom;
    meanp[eqVZ] = (rl * left[eqVZ] + rr * right[eqVZ]) * denom;
    meanp[eqHH] = (rl * lH + rr * rH) * denom;
    double v2_mean = meanp[eqVX] * meanp[eqVX] + meanp[eqVY] * meanp[eqVY] + meanp[eqVZ] * meanp[eqVZ];
    double a_mean = std::sqrt((g - 1.0) * std::max(meanp[eqHH] - 0.5 * v2_mean, 1.0e-12 * v2_mean));
    // set_eigenvalues :350-386
    eval[0] = meanp[eqVX] - a_mean;
    eval[1] = eval[2] = eval[3] = meanp[eqVX];
    eval[4] = meanp[eqVX] + a_mean;
    for (int v = 0; v < 5; v++) {
      if (eval[v] < 0.0) eval[v] = std::min(eval[v], -hc_eta);
      else eval[v] = std::max(eval[v], hc_eta);
    }
    // set_eigenvectors :395-437
    evec[0][eqRHO] = 1.0;
    evec[0][eqMMX] = meanp[eqVX] - a_mean;
    evec[0][eqMMY] = meanp[eqVY];
    evec[0][eqMMZ] = meanp[eqVZ];
    evec[0][eqERG] = meanp[eqHH] - meanp[eqVX] * a_mean;
    evec[1][eqRHO] = 1.0;
    evec[1][eqMMX] = meanp[eqVX];
    evec[1][eqMMY] = meanp[eqVY];
    evec[1][eqMMZ] = meanp[eqVZ];
    evec[1][eqERG] = 0.5 * v2_mean;
    evec[2][eqRHO] = 0.0; evec[2][eqMMX] = 0.0; evec[2][eqMMY] = 1.0; evec[2][eqMMZ] = 0.0;
    evec[2][eqERG] = meanp[eqVY];
    evec[3][eqRHO] = 0.0; evec[3][eqMMX] = 0.0; evec[3][eqMMY] = 0.0; evec[3][eqMMZ] = 1.0;
    evec[3][eqERG] = meanp[eqVZ];
    evec[4][eqRHO] = 1.0;
    evec[4][eqMMX] = meanp[eqVX] + a_mean;
    evec[4][eqMMY] = meanp[eqVY];
    evec[4][eqMMZ] = meanp[eqVZ];
    evec[4][eqERG] = meanp[eqHH] + meanp[eqVX] * a_mean;
    // set_ul_ur_udiff :446-476
    euler_PtoU(left, ul, g);
    euler_PtoU(right, ur, g);
    for (int v = 0; v < 5; v++) {
      if (equalD(ur[v], ul[v])) udiff[v] = 0.0;
      else udiff[v] = ur[v] - ul[v];
    }
    // set_wave_strengths :484-512
    strength[2] = udiff[eqMMY] - meanp[eqVY] * udiff[eqRO];
    strength[3] = udiff[eqMMZ] - meanp[eqVZ] * udiff[eqRO];
    double u5bar = udiff[eqERG] - strength[2] * meanp[eqVY] - strength[3] * meanp[eqVZ];
    strength[1] = (udiff[eqRHO] * (meanp[eqHH] - meanp[eqVX] * meanp[eqVX]) +
                   meanp[eqVX] * udiff[eqMMX] - u5bar) *
                  (g - 1.0) / a_mean / a_mean;
    strength[0] = 0.5 *
                  (udiff[eqRHO] * (meanp[eqVX] + a_mean) - udiff[eqMMX] - a_mean * strength[1]) /
                  a_mean;
    strength[4] = udiff[eqRHO] - strength[0] - strength[1];
    // calculate_symmetric_flux :520-565
    euler_UtoFlux(ul, out_flux, g);
    double fr[5];
    euler_UtoFlux(ur, fr, g);
    for (int v = 0; v < 5; v++) out_flux[v] += fr[v];
    for (int iw = 0; iw < 5; iw++) {
      out_flux[eqRHO] -= strength[iw] * std::fabs(eval[iw]) * evec[iw][eqRHO];
      out_flux[eqMMX] -= strength[iw] * std::fabs(eval[iw]) * evec[iw][eqMMX];
      out_flux[eqMMY] -= strength[iw] * std::fabs(eval[iw]) * evec[iw][eqMMY];
      out_flux[eqMMZ] -= strength[iw] * std::fabs(eval[iw]) * evec[iw][eqMMZ];
      out_flux[eqERG] -= strength[iw] * std::fabs(eval[iw]) * evec[iw][eqERG];
    }
    for (int v = 0; v < 5; v++) out_flux[v] *= 0.5;
    // set_pstar_from_meanp :573-597
    for (int v = 0; v < 5; v++) out_pstar[v] = meanp[v];
    out_pstar[eqPG] = meanp[eqRO] * a_mean * a_mean / g;
    return 0;
  }

  // ---------------------------------------------------------------------
  // Roe primitive-variable solver: Roe_Hydro_PrimitiveVar_solver.cpp:57-209
  int Roe_prim_var_solver(const double *l, const double *r, const double g, double *pstar)
  {
    const int eqHH = eqPG;
    double rl = std::sqrt(l[eqRO]), rr = std::sqrt(r[eqRO]), lH = Enthalpy(l, g),
           rH = Enthalpy(r, g), denom = 1.0 / (rl + rr), a_mean = 0.0, v2_mean = 0.0;
    double meanp[5];
    meanp[eqRO] = rl * rr;
    meanp[eqVX] = (rl * l[eqVX] + rr * r[eqVX]) * denom;
    meanp[eqVY] = (rl * l[eqVY] + rr * r[eqVY]) * denom;
    meanp[eqVZ] = (rl * l[eqVZ] + rr * r[eqVZ]) * denom;
    meanp[eqHH] = (rl * lH + rr * rH) * denom;
    v2_mean = meanp[eqVX] * meanp[eqVX] + meanp[eqVY] * meanp[eqVY] + meanp[eqVZ] * meanp[eqVZ];
    a_mean = std::sqrt((g - 1.0) * (meanp[eqHH] - 0.5 * v2_mean));
    meanp[eqPG] = meanp[eqRO] * a_mean * a_mean / g;
    if (meanp[eqVX] - a_mean >= 0.) {
      for (int i = 0; i < 5; i++) pstar[i] = l[i];
    }
    else if (meanp[eqVX] + a_mean <= 0.) {
      for (int i = 0; i < 5; i++) pstar[i] = r[i];
    }
    else {
      pstar[eqPG] = 0.5 * (l[eqPG] + r[eqPG] - meanp[eqRO] * a_mean * (r[eqVX] - l[eqVX]));
      pstar[eqVX] = 0.5 * (l[eqVX] + r[eqVX] - (r[eqPG] - l[eqPG]) / meanp[eqRO] / a_mean);
      if (pstar[eqVX] > 0.0) {
        pstar[eqRO] = l[eqRO] + meanp[eqRO] * (l[eqVX] - pstar[eqVX]) / a_mean;
      }
      else {
        pstar[eqRO] = r[eqRO] + meanp[eqRO] * (pstar[eqVX] - r[eqVX]) / a_mean;
      }
      if (pstar[eqVX] > 0.0) {
        pstar[eqVY] = l[eqVY];
        pstar[eqVZ] = l[eqVZ];
      }
      else {
        pstar[eqVY] = r[eqVY];
        pstar[eqVZ] = r[eqVZ];
      }
    }
    return 0;
  }

  // ---------------------------------------------------------------------
  // van Leer flux-vector splitting: Riemann_FVS_hydro.cpp:83-195, 204-240
  int FVS_flux(const double *pl, const double *pr, double *flux, double *pstar)
  {
    const double g = gamma;
    double fpos[5], fneg[5];
    double cl_ = chydro(pl, g), cr_ = chydro(pr, g), Ml = pl[eqVX] / cl_, Mr = pr[eqVX] / cr_,
           f1 = 0.0, f2 = 0.0;
    if (Ml < -1.0) {
      for (int v = 0; v < 5; v++) fpos[v] = 0.0;
    }
    else if (Ml > 1.0) {
      double utemp[5];
      euler_PtoU(pl, utemp, g);
      euler_PUtoFlux(pl, utemp, fpos);
    }
    else {
      f1 = 0.25 * pl[eqRO] * cl_ * (1.0 + Ml) * (1.0 + Ml);
      f2 = cl_ * ((g - 1.0) * Ml + 2);
      fpos[eqRHO] = f1;
      fpos[eqMMX] = f1 * f2 / g;
      fpos[eqMMY] = f1 * pl[eqVY];
      fpos[eqMMZ] = f1 * pl[eqVZ];
      fpos[eqERG] = f1 * (f2 * f2 * 0.5 / (g * g - 1.0) +
                          0.5 * (pl[eqVY] * pl[eqVY] + pl[eqVZ] * pl[eqVZ]));
    }
    if (Mr > 1.0) {
      for (int v = 0; v < 5; v++) fneg[v] = 0.0;
    }
    else if (Mr < -1.0) {
      double utemp[5];
      euler_PtoU(pr, utemp, g);
      euler_PUtoFlux(pr, utemp, fneg);
    }
    else {
      f1 = -0.25 * pr[eqRO] * cr_ * (1.0 - Mr) * (1.0 - Mr);
      f2 = cr_ * ((g - 1.0) * Mr - 2);
      fneg[eqRHO] = f1;
      fneg[eqMMX] = f1 * f2 / g;
      fneg[eqMMY] = f1 * pr[eqVY];
      fneg[eqMMZ] = f1 * pr[eqVZ];
      fneg[eqERG] = f1 * (f2 * f2 * 0.5 / (g * g - 1) +
                          0.5 * (pr[eqVY] * pr[eqVY] + pr[eqVZ] * pr[eqVZ]));
    }
    for (int v = 0; v < 5; v++) flux[v] = fpos[v] + fneg[v];
    // Roe_average_state :204-240
    double RoeAvg_rl = std::sqrt(pl[eqRO]), RoeAvg_rr = std::sqrt(pr[eqRO]),
           RoeAvg_denom = 1.0 / (RoeAvg_rl + RoeAvg_rr);
    double *ans = pstar;
    ans[eqRO] = RoeAvg_rl * RoeAvg_rr;
    ans[eqVX] = (RoeAvg_rl * pl[eqVX] + RoeAvg_rr * pr[eqVX]) * RoeAvg_denom;
    ans[eqVY] = (RoeAvg_rl * pl[eqVY] + RoeAvg_rr * pr[eqVY]) * RoeAvg_denom;
    ans[eqVZ] = (RoeAvg_rl * pl[eqVZ] + RoeAvg_rr * pr[eqVZ]) * RoeAvg_denom;
    ans[eqPG] = RoeAvg_denom * (RoeAvg_rl * Enthalpy(pl, g) + RoeAvg_rr * Enthalpy(pr, g));
    ans[eqPG] = (g - 1.0) * (ans[eqPG] - 0.5 * (ans[eqVX] * ans[eqVX] + ans[eqVY] * ans[eqVY] +
                                                ans[eqVZ] * ans[eqVZ]));
    ans[eqPG] = ans[eqRO] * ans[eqPG] / g;
    return 0;
  }

  // ---------------------------------------------------------------------
  // HLL (hydro): HLL_hydro.cpp:92-164.  PtoU / PUtoFlux are the virtual
  // (tracer-carrying) versions; HD_FL/HD_FR persist between calls.
  int hydro_HLL_flux_solver(const double *Pl, const double *Pr, const double g, double *out_flux,
                            double *out_ustar)
  {
    double HD_UL[MAXNV], HD_UR[MAXNV];
    PtoU(Pl, HD_UL, g);
    PtoU(Pr, HD_UR, g);
    PUtoFlux(Pl, HD_UL, HD_FL);
    PUtoFlux(Pr, HD_UR, HD_FR);
    double cf_l = chydro(Pl, g), cf_r = chydro(Pr, g);
    double cf_max = std::max(cf_l, cf_r);
    double Sl = std::min(Pl[eqVX], Pr[eqVX]) - cf_max;
    double Sr = std::max(Pl[eqVX], Pr[eqVX]) + cf_max;
    if (Sl > 0) {
      for (int v = 0; v < nvar; v++) out_flux[v] = HD_FL[v];
    }
    else if (Sr < 0) {
      for (int v = 0; v < nvar; v++) out_flux[v] = HD_FR[v];
    }
    else {
      for (int v = 0; v < nvar; v++)
        out_flux[v] = (Sr * HD_FL[v] - Sl * HD_FR[v] + Sr * Sl * (HD_UR[v] - HD_UL[v])) / (Sr - Sl);
    }
    for (int v = 0; v < nvar; v++)
      out_ustar[v] = (Sr * HD_UR[v] - Sl * HD_UL[v] + HD_FL[v] - HD_FR[v]) / (Sr - Sl);
    return 0;
  }

  // ---------------------------------------------------------------------
  // Linear / exact / hybrid hydro Riemann solver: riemann.cpp:245-464 with
  // linear_solver :674-748, linearOK :612-621, exact_solver :754-823,
  // solve_rarerare :829-885, solve_cavitation :891-963, check_wave_locations
  // :471-586, HydroWave(Full) eqns_hydro_adiabatic.cpp:221-300 and the Brent
  // root finder findroot.cpp:158-181, 270-309, 359-452.
  int HydroWave(int lr, const double pp, const double *prewave, double *u, const double g) const
  {
    double pratio = pp / prewave[eqPG];
    double c0 = std::sqrt(g * prewave[eqPG] / prewave[eqRO]);
    if (pratio < 1) {
      *u = 2. * c0 / (g - 1.) * (1 - std::exp((g - 1.) / 2. / g * std::log(pratio)));
      if (lr == XN) *u = prewave[eqVX] + (*u);
      else *u = prewave[eqVX] - (*u);
    }
    else if (pratio > 1) {
      *u = c0 * (pratio - 1.) / std::sqrt(g * (g - 1.) / 2. * (1. + pratio * (g + 1.) / (g - 1.)));
      if (lr == XN) *u = prewave[eqVX] - (*u);
      else *u = prewave[eqVX] + (*u);
    }
    else {
      *u = prewave[eqVX];
    }
    return 0;
  }
  int HydroWaveFull(int lr, const double pp, const double *prewave, double *u, double *rho,
                    const double g) const
  {
    double pratio = pp / prewave[eqPG];
    HydroWave(lr, pp, prewave, u, g);
    if (pratio < 1) *rho = prewave[eqRO] * std::exp(std::log(pratio) / g);
    else if (pratio > 1)
      *rho = prewave[eqRO] * (1 + pratio * (g + 1) / (g - 1.)) / ((g + 1.) / (g - 1.) + pratio);
    else *rho = prewave[eqRO];
    return 0;
  }
  double FR_root_function(double pp) const
  {
    double ustarL, ustarR;
    HydroWave(XN, pp, rs_left, &ustarL, gamma);
    HydroWave(XP, pp, rs_right, &ustarR, gamma);
    return (ustarR - ustarL);
  }
  int bracket_root_pos(double *x1, double *x2) const
  {
    const float factor = 1.6f;  // `float factor=1.6;` in the reference (findroot.cpp:272)
    if (*x1 == *x2) return 1;
    if (*x1 > *x2) std::swap(*x1, *x2);
    double f1 = FR_root_function(*x1);
    double f2 = FR_root_function(*x2);
    for (int j = 0; j < 50; j++) {
      if (f1 * f2 < 0) return 0;
      if (std::fabs(f1) < std::fabs(f2)) f1 = FR_root_function(*x1 *= 1. / factor);
      else f2 = FR_root_function(*x2 *= factor);
    }
    f1 = FR_root_function(*x1 = 0.);
    if (f1 * f2 < 0) return 0;
    *x1 = *x2 = 0.;
    return 1;
  }
  int find_root_zbrent(double x1, double x2, double tol, double *ans) const
  {
    const int ITMAX = 100;
    const double EPS = MACHINEACCURACY;
    double a = x1, b = x2, c = x2, d, e, min1, min2;
    double fa = FR_root_function(a), fb = FR_root_function(b), fc, p, q, r, s, tol1, xm;
    d = 0.;
    e = 0.;
    if ((fa > 0.0 && fb > 0.0) || (fa < 0.0 && fb < 0.0)) return 1;
    fc = fb;
    for (int iter = 1; iter <= ITMAX; iter++) {
      if ((fb > 0.0 && fc > 0.0) || (fb < 0.0 && fc < 0.0)) {
        c = a;
        fc = fa;
        e = d = b - a;
      }
      if (std::fabs(fc) < std::fabs(fb)) {
        a = b; b = c; c = a;
        fa = fb; fb = fc; fc = fa;
      }
      tol1 = 2.0 * EPS * std::fabs(b) + 0.5 * tol * std::fabs(b);
      xm = 0.5 * (c - b);
      if (std::fabs(xm) <= tol1 || fb == 0.0) {
        *ans = b;
        return 0;
      }
      if (std::fabs(e) >= tol1 && std::fabs(fa) > std::fabs(fb)) {
        s = fb / fa;
        if (a == c) {
          p = 2.0 * xm * s;
          q = 1.0 - s;
        }
        else {
          q = fa / fc;
          r = fb / fc;
          p = s * (2.0 * xm * q * (q - r) - (b - a) * (r - 1.0));
          q = (q - 1.0) * (r - 1.0) * (s - 1.0);
        }
        if (p > 0.0) q = -q;
        p = std::fabs(p);
        min1 = 3.0 * xm * q - std::fabs(tol1 * q);
        min2 = std::fabs(e * q);
        if (2.0 * p < (min1 < min2 ? min1 : min2)) {
          e = d;
          d = p / q;
        }
        else {
          d = xm;
          e = d;
        }
      }
      else {
        d = xm;
        e = d;
      }
      a = b;
      fa = fb;
      if (std::fabs(d) > tol1) b += d;
      else b += ((xm) >= 0.0 ? std::fabs(tol1) : -std::fabs(tol1));
      fb = FR_root_function(b);
    }
    return 1;
  }
  int FR_find_root(double *ans, const double p1, const double p2) const
  {
    double x1 = (p1 + p2) / 6.0;
    double x2 = x1 * 9.0;
    if (bracket_root_pos(&x1, &x2) != 0) {
      *ans = -1.0;
      return 1;
    }
    if (find_root_zbrent(x1, x2, 1.0e-8, ans) != 0) {
      *ans = -1.0;
      return 1;
    }
    return 0;
  }
  int check_wave_locations()
  {
    const double g = gamma;
    if (rs_pstar[eqPG] < rs_left[eqPG]) {
      if (rs_left[eqVX] >= cl) {
        rs_pstar[eqPG] = rs_left[eqPG];
        rs_pstar[eqRO] = rs_left[eqRO];
        rs_pstar[eqVX] = rs_left[eqVX];
        return 0;
      }
      else if (rs_pstar[eqVX] > 0.) {
        double cstar = chydro(rs_pstar, g);
        if (rs_pstar[eqVX] > cstar) {
          rs_pstar[eqVX] = (2. * cl + rs_left[eqVX] * (g - 1.)) / (g + 1.);
          rs_pstar[eqRO] = rs_left[eqRO] * std::exp(2. / (g - 1.) * std::log(rs_pstar[eqVX] / cl));
          rs_pstar[eqPG] = std::exp(g * std::log(rs_pstar[eqRO] / rs_left[eqRO])) * rs_left[eqPG];
          return 0;
        }
      }
    }
    if (rs_pstar[eqPG] < rs_right[eqPG]) {
      if (rs_right[eqVX] <= -cr) {
        rs_pstar[eqPG] = rs_right[eqPG];
        rs_pstar[eqRO] = rs_right[eqRO];
        rs_pstar[eqVX] = rs_right[eqVX];
        return 0;
      }
      else if (rs_pstar[eqVX] < 0.) {
        double cstar = chydro(rs_pstar, g);
        if (rs_pstar[eqVX] < -cstar) {
          rs_pstar[eqVX] = (-2. * cr + rs_right[eqVX] * (g - 1.)) / (g + 1.);
          rs_pstar[eqRO] = rs_right[eqRO] * std::exp(2. / (g - 1.) * std::log(-rs_pstar[eqVX] / cr));
          rs_pstar[eqPG] = std::exp(g * std::log(rs_pstar[eqRO] / rs_right[eqRO])) * rs_right[eqPG];
          return 0;
        }
      }
    }
    if (rs_pstar[eqPG] > 1.0000001 * rs_right[eqPG]) {
      double vsh = rs_right[eqVX] + (rs_pstar[eqPG] / rs_right[eqPG] - 1.) * cr * cr / g /
                                        (rs_pstar[eqVX] - rs_right[eqVX]);
      if (vsh < 0.) {
        rs_pstar[eqPG] = rs_right[eqPG];
        rs_pstar[eqRO] = rs_right[eqRO];
        rs_pstar[eqVX] = rs_right[eqVX];
        return 0;
      }
    }
    if (rs_pstar[eqPG] > 1.0000001 * rs_left[eqPG]) {
      double vsh = rs_left[eqVX] + (rs_pstar[eqPG] / rs_left[eqPG] - 1.) * cl * cl / g /
                                       (rs_pstar[eqVX] - rs_left[eqVX]);
      if (vsh > 0.) {
        rs_pstar[eqPG] = rs_left[eqPG];
        rs_pstar[eqRO] = rs_left[eqRO];
        rs_pstar[eqVX] = rs_left[eqVX];
        return 0;
      }
    }
    return 0;
  }
  int linearOK() const
  {
    if ((std::max(rs_left[eqPG], rs_right[eqPG]) / std::min(rs_left[eqPG], rs_right[eqPG]) < 1.4) &&
        (std::max(rs_left[eqRO], rs_right[eqRO]) / std::min(rs_left[eqRO], rs_right[eqRO]) < 1.4) &&
        (std::fabs(rs_right[eqVX] - rs_left[eqVX]) / std::min(cl, cr) < 0.03))
      return 0;
    return 1;
  }
  int linear_solver()
  {
    const double g = gamma;
    for (int i = 0; i < 5; i++) rs_meanp[i] = (rs_left[i] + rs_right[i]) / 2.;
    double mcs = chydro(rs_meanp, g);
    if (rs_meanp[eqVX] - mcs >= 0.) {
      for (int i = 0; i < 5; i++) rs_pstar[i] = rs_left[i];
      return 0;
    }
    else if (rs_meanp[eqVX] + mcs <= 0.) {
      for (int i = 0; i < 5; i++) rs_pstar[i] = rs_right[i];
      return 0;
    }
    else {
      rs_pstar[eqPG] =
          0.5 * (rs_left[eqPG] + rs_right[eqPG] - rs_meanp[eqRO] * mcs * (rs_right[eqVX] - rs_left[eqVX]));
      rs_pstar[eqVX] =
          0.5 * (rs_left[eqVX] + rs_right[eqVX] - (rs_right[eqPG] - rs_left[eqPG]) / rs_meanp[eqRO] / mcs);
      if (std::fabs(rs_pstar[eqVX] / mcs) <= 1.e-6) {
        rs_pstar[eqRO] = rs_meanp[eqRO] * (2. + (rs_left[eqVX] - rs_right[eqVX]) / mcs) / 2.;
      }
      else if (rs_pstar[eqVX] > 0) {
        rs_pstar[eqRO] = rs_left[eqRO] + rs_meanp[eqRO] * (rs_left[eqVX] - rs_pstar[eqVX]) / mcs;
      }
      else if (rs_pstar[eqVX] < 0) {
        rs_pstar[eqRO] = rs_right[eqRO] + rs_meanp[eqRO] * (rs_pstar[eqVX] - rs_right[eqVX]) / mcs;
      }
      else return 1;
    }
    return 0;
  }
  int exact_solver()
  {
    const double g = gamma;
    int err = 0;
    err += FR_find_root(&(rs_pstar[eqPG]), rs_left[eqPG], rs_right[eqPG]);
    err += HydroWaveFull(XN, rs_pstar[eqPG], rs_left, &(rs_pstar[eqVX]), &(rs_pstar[eqRO]), g);
    double rhostar, temp;
    if ((rs_pstar[eqVX] > 0) && (std::fabs(rs_pstar[eqVX] / cr) > 1.e-6)) {
      err += HydroWaveFull(XN, rs_pstar[eqPG], rs_left, &temp, &rhostar, g);
    }
    else if ((rs_pstar[eqVX] < 0) && (std::fabs(rs_pstar[eqVX] / cr) > 1.e-6)) {
      err += HydroWaveFull(XP, rs_pstar[eqPG], rs_right, &temp, &rhostar, g);
    }
    else if (std::fabs(rs_pstar[eqVX] / cr) <= 1.e-6) {
      err += HydroWaveFull(XN, rs_pstar[eqPG], rs_left, &temp, &rhostar, g);
      err += HydroWaveFull(XP, rs_pstar[eqPG], rs_right, &temp, &(rs_pstar[eqRO]), g);
      rhostar = (rhostar + rs_pstar[eqRO]) / 2.0;
    }
    else {
      rs_pstar[eqRO] = -1.0;
      return 1;
    }
    rs_pstar[eqRO] = rhostar;
    if (err != 0) {
      rs_pstar[eqPG] = rs_pstar[eqRO] = rs_pstar[eqVX] = -1.9;
      return 1;
    }
    check_wave_locations();
    return 0;
  }
  int solve_rarerare()
  {
    const double g = gamma;
    rs_pstar[eqPG] =
        std::pow((cl + cr - (g - 1.) / 2. * (rs_right[eqVX] - rs_left[eqVX])) /
                     ((cl * std::exp(-(g - 1.) / 2. / g * std::log(rs_left[eqPG]))) +
                      (cr * std::exp(-(g - 1.) / 2. / g * std::log(rs_right[eqPG])))),
                 2. * g / (g - 1.));
    rs_pstar[eqVX] = rs_left[eqVX] +
                     2. * cl / (g - 1.) *
                         (1. - std::exp((g - 1.) / 2. / g * std::log(rs_pstar[eqPG] / rs_left[eqPG])));
    if ((rs_pstar[eqVX] > 0) && (std::fabs(rs_pstar[eqVX] / cr) > 1.e-6)) {
      rs_pstar[eqRO] = rs_left[eqRO] * std::exp(std::log(rs_pstar[eqPG] / rs_left[eqPG]) / g);
    }
    else if ((rs_pstar[eqVX] < 0) && (std::fabs(rs_pstar[eqVX] / cr) > 1.e-6)) {
      rs_pstar[eqRO] = rs_right[eqRO] * std::exp(std::log(rs_pstar[eqPG] / rs_right[eqPG]) / g);
    }
    else if (std::fabs(rs_pstar[eqVX] / cr) <= 1.e-6) {
      rs_pstar[eqRO] = ((rs_right[eqRO] * std::exp(std::log(rs_pstar[eqPG] / rs_right[eqPG]) / g)) +
                        (rs_left[eqRO] * std::exp(std::log(rs_pstar[eqPG] / rs_left[eqPG]) / g))) /
                       2.0;
    }
    else {
      rs_pstar[eqRO] = -1.0;
      return 1;
    }
    check_wave_locations();
    return 0;
  }
  int solve_cavitation()
  {
    const double g = gamma;
    if ((rs_left[eqVX] - cl) >= 0.) {
      for (int i = 0; i < 5; i++) rs_pstar[i] = rs_left[i];
      return 0;
    }
    double temp = 2. / (g - 1.);
    if ((rs_left[eqVX] + temp * cl) >= 0.) {
      rs_pstar[eqVX] = (2. * cl + rs_left[eqVX] * (g - 1.)) / (g + 1.);
      rs_pstar[eqRO] = rs_left[eqRO] * std::exp(2. / (g - 1.) * std::log(rs_pstar[eqVX] / cl));
      rs_pstar[eqPG] = std::exp(g * std::log(rs_pstar[eqRO] / rs_left[eqRO])) * rs_left[eqPG];
      return 0;
    }
    if ((rs_right[eqVX] - temp * cr) >= 0.) {
      rs_pstar[eqRO] = refvec[eqRO] * BASEPG;
      rs_pstar[eqPG] = refvec[eqPG] * BASEPG;
      rs_pstar[eqVX] = refvec[eqVX] * BASEPG;
      return 0;
    }
    if ((rs_right[eqVX] + cr) > 0.) {
      rs_pstar[eqVX] = (-2. * cr + rs_right[eqVX] * (g - 1.)) / (g + 1.);
      rs_pstar[eqRO] = rs_right[eqRO] * std::exp(2. / (g - 1.) * std::log(-rs_pstar[eqVX] / cr));
      rs_pstar[eqPG] = std::exp(g * std::log(rs_pstar[eqRO] / rs_right[eqRO])) * rs_right[eqPG];
      return 0;
    }
    if ((rs_right[eqVX] + cr) <= 0.) {
      for (int i = 0; i < 5; i++) rs_pstar[i] = rs_right[i];
      return 0;
    }
    return 1;
  }
  int JMs_riemann_solve(const double *l, const double *r, double *ans, const int mode, const double g)
  {
    int err = 0;
    gamma = g;
    if (l[eqRO] < TINYVALUE || l[eqPG] < TINYVALUE || r[eqRO] < TINYVALUE || r[eqPG] < TINYVALUE)
      throw physics_error("riemann::solve() Density/Pressure too small");
    for (int v = 0; v < 5; v++) rs_left[v] = l[v];
    for (int v = 0; v < 5; v++) rs_right[v] = r[v];
    double diff = 0.;
    for (int i = 0; i < 5; i++)
      diff += std::fabs(rs_right[i] - rs_left[i]) / (std::fabs(refvec[i]) + TINYVALUE);
    if (diff < 1.e-6) {
      for (int i = 0; i < 5; i++) {
        rs_pstar[i] = (l[i] + r[i]) / 2.;
        ans[i] = rs_pstar[i];
      }
      return 0;
    }
    cl = chydro(rs_left, g);
    cr = chydro(rs_right, g);
    if ((rs_right[eqVX] - rs_left[eqVX]) <= 2. * (cl + std::sqrt((g - 1.) / 2. / g) * cr) / (g - 1.)) {
      switch (mode) {
        case FLUX_RSlinear:
          err = linear_solver();
          if (err != 0) {
            rs_pstar[eqPG] = rs_pstar[eqRO] = rs_pstar[eqVX] = TINYVALUE;
            for (int i = 0; i < 5; i++) ans[i] = rs_pstar[i];
            return 1;
          }
          break;
        case FLUX_RSexact:
          err = exact_solver();
          if (err != 0) {
            rs_pstar[eqPG] = rs_pstar[eqRO] = TINYVALUE;
            for (int i = 0; i < 5; i++) ans[i] = rs_pstar[i];
            return 1;
          }
          break;
        case FLUX_RShybrid:
          err = linear_solver();
          if (err != 0) rs_pstar[eqPG] = rs_pstar[eqRO] = rs_pstar[eqVX] = TINYVALUE;
          if (err != 0 || linearOK() != 0) {
            err = exact_solver();
            if (err != 0) {
              rs_pstar[eqPG] = rs_pstar[eqRO] = TINYVALUE;
              for (int i = 0; i < 5; i++) ans[i] = rs_pstar[i];
              return err;
            }
          }
          break;
        default:
          throw physics_error("switch not known in RiemannEul::solve()");
      }
    }
    else if ((rs_right[eqVX] - rs_left[eqVX]) <= 2. * (cl + cr) / (g - 1.)) {
      err = solve_rarerare();
      if (err != 0) {
        rs_pstar[eqPG] = rs_pstar[eqRO] = rs_pstar[eqVX] = -1.9e99;
        for (int i = 0; i < 5; i++) ans[i] = rs_pstar[i];
        return err;
      }
    }
    else {
      err = solve_cavitation();
      if (err) {
        rs_pstar[eqPG] = rs_pstar[eqRO] = rs_pstar[eqVX] = -1.9e100;
        for (int i = 0; i < 5; i++) ans[i] = rs_pstar[i];
        return err;
      }
    }
    if (rs_pstar[eqVX] > 0) {
      rs_pstar[eqVY] = rs_left[eqVY];
      rs_pstar[eqVZ] = rs_left[eqVZ];
    }
    else {
      rs_pstar[eqVY] = rs_right[eqVY];
      rs_pstar[eqVZ] = rs_right[eqVZ];
    }
    if (rs_pstar[eqPG] <= TINYVALUE) rs_pstar[eqPG] = BASEPG * refvec[eqPG];
    if (rs_pstar[eqRO] <= TINYVALUE) rs_pstar[eqRO] = BASEPG * refvec[eqRO];
    for (int i = 0; i < 5; i++) ans[i] = rs_pstar[i];
    return 0;
  }

  // ---------------------------------------------------------------------
  // FKJ98 linear MHD Riemann solver with Roe-Balsara eigenvectors
  // (riemann_MHD::JMs_riemann_solve, Riemann_solvers/riemannMHD.cpp:165-405; speeds :555-766,
  // eigenvalues :776-794, eigenvectors :965-1117, strengths :816-845, P* :849-963).
  // Solver-frame ordering (riemannMHD.h:56-65): 0 rho, 1 p, 2 vx, 3 vy, 4 vz, 5 By, 6 Bz, 7 Bx.
  // Wave ordering (riemannMHD.h:34-42): F-, A-, S-, contact, S+, A+, F+.
  // The reference's rep.error() exits are thrown as physics_error.
  int mhd_JMs_riemann_solve(const double *l, const double *r, double *ans, const int mode, const double g)
  {
    if (mode != FLUX_RSlinear) throw physics_error("riemann_MHD: MODE i: Don't know what to do.");
    gamma = g;
    const int map[8] = {eqRO, eqPG, eqVX, eqVY, eqVZ, eqBY, eqBZ, eqBX};  // code2solvervars :426-452
    double L[8], R[8], M[8], star[8];
    for (int i = 0; i < 8; i++) {
      L[i] = l[map[i]];
      R[i] = r[map[i]];
    }
    for (int i = 0; i < 8; i++) M[i] = 0.5 * (L[i] + R[i]);  // get_average_state :534-546
    const double bxs = M[7];
    star[7] = bxs;
    double diff = 0.;
    for (int i = 0; i < 7; i++) diff += std::fabs(R[i] - L[i]) / (std::fabs(refvec[i]) + TINYVALUE);
    auto put_back = [&](const double *sv) {  // solver2codevars :480-509; entries > 7 stay 0
      for (int v = 0; v < nvar; v++) ans[v] = 0.0;
      for (int i = 0; i < 8; i++) ans[map[i]] = sv[i];
    };
    if (diff < 1.e-6) {
      for (int i = 0; i < 7; i++) star[i] = M[i];
      put_back(star);
      return 0;
    }
    const double smallB = MACHINEACCURACY, tinyB = smallB * smallB * smallB;
    // get_sound_speeds :555-766
    const double ch = std::sqrt(g * M[1] / M[0]);
    const double bx = bxs / std::sqrt(M[0]);
    double ca = std::fabs(bx);
    const double bt = std::sqrt((M[5] * M[5] + M[6] * M[6]) / M[0]);
    double betay, betaz;
    if (bt > tinyB) {
      betay = M[5] / std::sqrt(M[0]) / bt;
      betaz = M[6] / std::sqrt(M[0]) / bt;
    }
    else {
      betay = 1. / std::sqrt(2.);
      betaz = 1. / std::sqrt(2.);
    }
    if ((ch / std::max(ca, bt)) < std::sqrt(smallB))
      throw physics_error("riemann_MHD::(get_sound_speeds) returned with error");
    double t1 = ch * ch + bx * bx + bt * bt;
    double t2 = 4. * ch * ch * bx * bx;
    if ((t2 = t1 * t1 - t2) < MACHINEACCURACY) t2 = MACHINEACCURACY;
    double cf = std::sqrt((t1 + std::sqrt(t2)) / 2.);
    if ((t2 = t1 - std::sqrt(t2)) < MACHINEACCURACY) t2 = MACHINEACCURACY;
    double cs = std::sqrt(t2 / 2.);
    if (cs > ch) cs = ch - smallB;
    if (ch > cf) cf = ch + smallB;
    if (cs > ca) cs = ca - smallB;
    if (cs <= 0. || cs > ca) cs = ca / 2.;
    if (ca > cf) cf = ca + smallB;
    double alphaf, alphas, cf2diff;
    if ((cf2diff = cf * cf - cs * cs) > smallB) {
      if ((alphaf = ch * ch - cs * cs) <= smallB) alphaf = 0.;
      if ((alphas = cf * cf - ch * ch) <= smallB) alphas = 0.;
      if ((alphaf = std::sqrt(alphaf / cf2diff)) > 1.) alphaf = 1.;
      if ((alphas = std::sqrt(alphas / cf2diff)) > 1.) alphas = 1.;
    }
    else throw physics_error("riemann_MHD: near triple degeneracy point (Bugging out for now...)");
    if ((cf <= 0.) || (cs < 0.) || (ca < 0.) || (ch <= 0.))
      throw physics_error("riemann_MHD::(get_sound_speeds) returned with error");
    // get_eigenvalues :776-794
    const double ev[7] = {M[2] - cf, M[2] - ca, M[2] - cs, M[2], M[2] + cs, M[2] + ca, M[2] + cf};
    // RoeBalsara_evectors :965-1117 (rows F-, A-, S-, C, S+, A+, F+; columns in solver ordering)
    const double r2 = std::sqrt(2.);
    const int sBx = (bxs < 0.) ? -1 : 1;
    double le[7][7], re[7][7];
    for (int w = 0; w < 7; w++)
      for (int j = 0; j < 7; j++) le[w][j] = re[w][j] = 0.0;
    const double sr0 = std::sqrt(M[0]);
    le[0][2] = -alphaf * cf;
    le[0][3] = alphas * cs * sBx * betay;
    le[0][4] = alphas * cs * sBx * betaz;
    le[0][1] = alphaf / M[0];
    le[0][5] = alphas * ch * betay / sr0;
    le[0][6] = alphas * ch * betaz / sr0;
    le[1][3] = sBx * betaz / r2;
    le[1][4] = -sBx * betay / r2;
    le[1][5] = betaz / sr0 / r2;
    le[1][6] = -betay / sr0 / r2;
    le[2][2] = -alphas * cs;
    le[2][3] = -alphaf * cf * sBx * betay;
    le[2][4] = -alphaf * cf * sBx * betaz;
    le[2][1] = alphas / M[0];
    le[2][5] = -alphaf * ch * betay / sr0;
    le[2][6] = -alphaf * ch * betaz / sr0;
    le[3][0] = 1.;
    le[3][1] = -1 / ch / ch;
    // positive-going waves mirror the negative ones: velocities flip for S and F, fields flip for A
    for (int pair = 0; pair < 3; pair++) {
      const int n = pair, q = 6 - pair;
      const double sv = (pair == 1) ? 1.0 : -1.0, sb = (pair == 1) ? -1.0 : 1.0;
      le[q][2] = sv * le[n][2];
      le[q][3] = sv * le[n][3];
      le[q][4] = sv * le[n][4];
      le[q][1] = le[n][1];
      le[q][5] = sb * le[n][5];
      le[q][6] = sb * le[n][6];
    }
    // the reference writes "-x" for the mirrored entries; sv*x with sv = -1.0 is the same bits,
    // and for the Alfven pair the velocity entries are copied (sv = +1) and the fields negated
    re[0][0] = alphaf * M[0];
    re[0][2] = le[0][2];
    re[0][3] = le[0][3];
    re[0][4] = le[0][4];
    re[0][1] = alphaf * M[0] * ch * ch;
    re[0][5] = le[0][5] * M[0];
    re[0][6] = le[0][6] * M[0];
    re[1][3] = le[1][3];
    re[1][4] = le[1][4];
    re[1][5] = le[1][5] * M[0];
    re[1][6] = le[1][6] * M[0];
    re[2][0] = alphas * M[0];
    re[2][2] = le[2][2];
    re[2][3] = le[2][3];
    re[2][4] = le[2][4];
    re[2][1] = alphas * M[0] * ch * ch;
    re[2][5] = le[2][5] * M[0];
    re[2][6] = le[2][6] * M[0];
    re[3][0] = 1.0;
    for (int pair = 0; pair < 3; pair++) {
      const int n = pair, q = 6 - pair;
      const double sv = (pair == 1) ? 1.0 : -1.0, sb = (pair == 1) ? -1.0 : 1.0;
      re[q][0] = re[n][0];
      re[q][2] = sv * re[n][2];
      re[q][3] = sv * re[n][3];
      re[q][4] = sv * re[n][4];
      re[q][1] = re[n][1];
      re[q][5] = sb * re[n][5];
      re[q][6] = sb * re[n][6];
    }
    const double a22 = 1. / (2. * ch * ch);
    for (int j = 0; j < 7; j++) {
      le[0][j] *= a22;
      le[2][j] *= a22;
      le[4][j] *= a22;
      le[6][j] *= a22;
    }
    // calculate_wave_strengths :816-845 (dot product accumulates from 0.0 over the 7 entries in order)
    double pd[7], str[7];
    for (int j = 0; j < 7; j++) pd[j] = R[j] - L[j];
    for (int w = 0; w < 7; w++) {
      double t = 0.0;
      for (int j = 0; j < 7; j++) t += le[w][j] * pd[j];
      str[w] = t;
    }
    // get_pstar :849-963
    {
      int i = 0;
      for (int j = 0; j < 7; j++) star[j] = L[j];
      while ((i < 7) && (ev[i] < 0.)) {
        for (int j = 0; j < 7; j++) star[j] += str[i] * re[i][j];
        i++;
      }
      if (std::fabs(M[2]) < (1.e-4 * ch)) {
        i = 6;
        for (int j = 0; j < 7; j++) pd[j] = R[j];
        while ((i >= 0) && (ev[i] > 0.)) {
          for (int j = 0; j < 7; j++) pd[j] -= str[i] * re[i][j];
          i--;
        }
        for (int v = 0; v < 7; v++) star[v] = 0.5 * (star[v] + pd[v]);
      }
    }
    if (star[1] < 0.) star[1] = refvec[1] * BASEPG;
    if (star[0] < 0.) star[0] = refvec[0] * BASEPG;
    put_back(star);
    return 0;
  }

  // ---------------------------------------------------------------------
  // Roe flux solver for ideal MHD in conserved variables, symmetric form (Cargo & Gallice 1997;
  // Riemann_Roe_MHD_CV::MHD_Roe_CV_flux_solver_symmetric, Roe_MHD_ConservedVar_solver.cpp:218-264):
  // average :345-405, differences :417-462, speeds :473-551, eigenvalues + H-correction :563-607,
  // strengths :615-686, right eigenvectors :699-821, flux :1074-1133, P* :299-331.
  // Every step returns 0 in the reference, so the FKJ98 fallback of inviscid_flux never triggers.
  int MHD_Roe_CV_flux_solver_symmetric(const double *left, const double *right, const double g,
                                       const double hc_etamax, double *out_pstar, double *out_flux)
  {
    gamma = g;
    double UL[MAXNV], UR[MAXNV];
    mhd_PtoU(left, UL, g);
    mhd_PtoU(right, UR, g);
    const int eqHH = eqPG;  // the mean state holds the enthalpy in the pressure slot
    auto enthalpy = [&](const double *p) {
      return ((p[eqRO] * (p[eqVX] * p[eqVX] + p[eqVY] * p[eqVY] + p[eqVZ] * p[eqVZ]) / 2.0 +
               (g * p[eqPG] / (g - 1.0)) + (p[eqBX] * p[eqBX] + p[eqBY] * p[eqBY] + p[eqBZ] * p[eqBZ])) /
              p[eqRO]);
    };
    double mp[MAXNV];
    for (int v = 0; v < nvar; v++) mp[v] = 0.0;  // Roe_meanp entries > 7 are never written
    const double rl = std::sqrt(left[eqRO]), rr = std::sqrt(right[eqRO]);
    const double lH = enthalpy(left), rH = enthalpy(right);
    const double denom = 1.0 / (rl + rr);
    mp[eqRO] = rl * rr;
    mp[eqVX] = (rl * left[eqVX] + rr * right[eqVX]) * denom;
    mp[eqVY] = (rl * left[eqVY] + rr * right[eqVY]) * denom;
    mp[eqVZ] = (rl * left[eqVZ] + rr * right[eqVZ]) * denom;
    mp[eqBY] = (rr * left[eqBY] + rl * right[eqBY]) * denom;
    mp[eqBZ] = (rr * left[eqBZ] + rl * right[eqBZ]) * denom;
    mp[eqBX] = 0.5 * (left[eqBX] + right[eqBX]);
    const int sgn = (mp[eqBX] >= 0.0) ? 1 : -1;
    mp[eqHH] = (rl * lH + rr * rH) * denom;
    const double V = std::sqrt(mp[eqVX] * mp[eqVX] + mp[eqVY] * mp[eqVY] + mp[eqVZ] * mp[eqVZ]);
    const double B = std::sqrt(mp[eqBX] * mp[eqBX] + mp[eqBY] * mp[eqBY] + mp[eqBZ] * mp[eqBZ]);
    const double Bt = std::sqrt(mp[eqBY] * mp[eqBY] + mp[eqBZ] * mp[eqBZ]);
    double by, bz;
    if (Bt >= TINYVALUE) {
      by = mp[eqBY] / Bt;
      bz = mp[eqBZ] / Bt;
    }
    else {
      by = 1.0 / std::sqrt(2.0);
      bz = 1.0 / std::sqrt(2.0);
    }
    // differences :417-462
    double ud[MAXNV], pd[MAXNV];
    for (int v = 0; v < nvar; v++) {
      ud[v] = UR[v] - UL[v];
      pd[v] = right[v] - left[v];
    }
    ud[eqBBX] = pd[eqBX] = 0.0;
    const double X = (pd[eqBY] * pd[eqBY] + pd[eqBZ] * pd[eqBZ]) * 0.5 * denom * denom;
    pd[eqPG] = ((0.5 * V * V - X) * pd[eqRO] -
                (mp[eqVX] * ud[eqMMX] + mp[eqVY] * ud[eqMMY] + mp[eqVZ] * ud[eqMMZ]) + ud[eqERG] -
                (mp[eqBY] * pd[eqBY] + mp[eqBZ] * pd[eqBZ])) *
               (g - 1.0);
    // wave speeds :473-551
    const double b2 = B * B / mp[eqRO];
    const double a = std::sqrt((2.0 - g) * X + (g - 1.0) * std::max((mp[eqHH] - 0.5 * V * V - b2), 1.0e-12 * V * V));
    const double astar2 = a * a + b2;
    double ca = std::sqrt(mp[eqBX] * mp[eqBX] / mp[eqRO]);
    double cs = astar2 * astar2 - 4.0 * a * a * ca * ca;
    if (cs <= 0.0) cs = 0.0;
    else cs = std::sqrt(cs);
    const double cf = std::sqrt(0.5 * (astar2 + cs));
    cs = astar2 - cs;
    if (cs <= 0.0) cs = 0.0;
    else cs = std::sqrt(0.5 * cs);
    if (ca > cf) ca = cf;
    if (cs > ca) cs = ca;
    double af, as, cf2diff;
    if ((cf2diff = cf * cf - cs * cs) > MACHINEACCURACY) {
      if ((af = a * a - cs * cs) < 0.0) af = 0.;
      if ((as = cf * cf - a * a) < 0.0) as = 0.;
      if ((af = std::sqrt(af / cf2diff)) > 1.0) af = 1.0;
      if ((as = std::sqrt(as / cf2diff)) > 1.0) as = 1.0;
    }
    else af = as = 1.0 / std::sqrt(2.0);
    // eigenvalues + H-correction :563-607
    double ev[7] = {mp[eqVX] - cf, mp[eqVX] - ca, mp[eqVX] - cs, mp[eqVX], mp[eqVX] + cs, mp[eqVX] + ca,
                    mp[eqVX] + cf};
    for (int v = 0; v < 7; v++) {
      if (ev[v] < 0.0) ev[v] = std::min(ev[v], -hc_etamax);
      else ev[v] = std::max(ev[v], hc_etamax);
    }
    // wave strengths (CG97 4.20) :615-686
    double st[7];
    const double ro = mp[eqRO], sro = std::sqrt(mp[eqRO]);
    st[0] = 0.5 * (af * (X * pd[eqRO] + pd[eqPG]) + ro * as * cs * sgn * (by * pd[eqVY] + bz * pd[eqVZ]) -
                   ro * af * cf * pd[eqVX] + sro * as * a * (by * pd[eqBY] + bz * pd[eqBZ]));
    st[6] = 0.5 * (af * (X * pd[eqRO] + pd[eqPG]) - ro * as * cs * sgn * (by * pd[eqVY] + bz * pd[eqVZ]) +
                   ro * af * cf * pd[eqVX] + sro * as * a * (by * pd[eqBY] + bz * pd[eqBZ]));
    st[2] = 0.5 * (as * (X * pd[eqRO] + pd[eqPG]) - ro * af * cf * sgn * (by * pd[eqVY] + bz * pd[eqVZ]) -
                   ro * as * cs * pd[eqVX] - sro * af * a * (by * pd[eqBY] + bz * pd[eqBZ]));
    st[4] = 0.5 * (as * (X * pd[eqRO] + pd[eqPG]) + ro * af * cf * sgn * (by * pd[eqVY] + bz * pd[eqVZ]) +
                   ro * as * cs * pd[eqVX] - sro * af * a * (by * pd[eqBY] + bz * pd[eqBZ]));
    st[1] = 0.5 * (+by * pd[eqVZ] - bz * pd[eqVY] + sgn * (by * pd[eqBZ] - bz * pd[eqBY]) / sro);
    st[5] = 0.5 * (-by * pd[eqVZ] + bz * pd[eqVY] + sgn * (by * pd[eqBZ] - bz * pd[eqBY]) / sro);
    st[3] = (a * a - X) * pd[eqRO] - pd[eqPG];
    // right eigenvectors :699-821; columns: rho, mx, my, mz, By, Bz, E
    double re[7][7];
    re[3][0] = 1;
    re[3][1] = mp[eqVX];
    re[3][2] = mp[eqVY];
    re[3][3] = mp[eqVZ];
    re[3][4] = 0.0;
    re[3][5] = 0.0;
    re[3][6] = 0.5 * V * V + X * (g - 2) / (g - 1);
    for (int v = 0; v < 7; v++) re[3][v] /= a * a;
    re[1][0] = 0.0;
    re[1][1] = 0.0;
    re[1][2] = -ro * bz;
    re[1][3] = +ro * by;
    re[1][4] = -sgn * sro * bz;
    re[1][5] = +sgn * sro * by;
    re[1][6] = -ro * (mp[eqVY] * bz - mp[eqVZ] * by);
    re[5][0] = 0.0;
    re[5][1] = 0.0;
    re[5][2] = -re[1][2];
    re[5][3] = -re[1][3];
    re[5][4] = re[1][4];
    re[5][5] = re[1][5];
    re[5][6] = -re[1][6];
    const double das = ro * as, daf = ro * af;
    re[2][0] = das;
    re[2][1] = das * (mp[eqVX] - cs);
    re[2][2] = das * mp[eqVY] - daf * cf * by * sgn;
    re[2][3] = das * mp[eqVZ] - daf * cf * bz * sgn;
    re[2][4] = -sro * af * a * by;
    re[2][5] = -sro * af * a * bz;
    re[2][6] = das * (mp[eqHH] - B * B / ro - mp[eqVX] * cs) - daf * cf * sgn * (mp[eqVY] * by + mp[eqVZ] * bz) -
               sro * af * a * Bt;
    re[4][0] = das;
    re[4][1] = das * (mp[eqVX] + cs);
    re[4][2] = das * mp[eqVY] + daf * cf * by * sgn;
    re[4][3] = das * mp[eqVZ] + daf * cf * bz * sgn;
    re[4][4] = re[2][4];
    re[4][5] = re[2][5];
    re[4][6] = das * (mp[eqHH] - B * B / ro + mp[eqVX] * cs) + daf * cf * sgn * (mp[eqVY] * by + mp[eqVZ] * bz) -
               sro * af * a * Bt;
    re[0][0] = daf;
    re[0][1] = daf * (mp[eqVX] - cf);
    re[0][2] = daf * mp[eqVY] + das * cs * by * sgn;
    re[0][3] = daf * mp[eqVZ] + das * cs * bz * sgn;
    re[0][4] = sro * as * a * by;
    re[0][5] = sro * as * a * bz;
    re[0][6] = daf * (mp[eqHH] - B * B / ro - mp[eqVX] * cf) + das * cs * sgn * (mp[eqVY] * by + mp[eqVZ] * bz) +
               sro * as * a * Bt;
    re[6][0] = daf;
    re[6][1] = daf * (mp[eqVX] + cf);
    re[6][2] = daf * mp[eqVY] - das * cs * by * sgn;
    re[6][3] = daf * mp[eqVZ] - das * cs * bz * sgn;
    re[6][4] = re[0][4];
    re[6][5] = re[0][5];
    re[6][6] = daf * (mp[eqHH] - B * B / ro + mp[eqVX] * cf) - das * cs * sgn * (mp[eqVY] * by + mp[eqVZ] * bz) +
               sro * as * a * Bt;
    const double norm = ro * a * a;
    for (int v = 0; v < 7; v++) re[2][v] /= norm;
    for (int v = 0; v < 7; v++) re[4][v] /= norm;
    for (int v = 0; v < 7; v++) re[0][v] /= norm;
    for (int v = 0; v < 7; v++) re[6][v] /= norm;
    // symmetric flux :1074-1133
    mhd_PUtoFlux(left, UL, out_flux);
    mhd_PUtoFlux(right, UR, UL);
    for (int v = 0; v < 8; v++) out_flux[v] += UL[v];
    const int col[7] = {eqRHO, eqMMX, eqMMY, eqMMZ, eqBBY, eqBBZ, eqERG};
    for (int w = 0; w < 7; w++)
      for (int c = 0; c < 7; c++) out_flux[col[c]] -= st[w] * std::fabs(ev[w]) * re[w][c];
    for (int v = 0; v < 8; v++) out_flux[v] *= 0.5;
    // set_pstar_from_meanp :299-331
    for (int v = 0; v < nvar; v++) out_pstar[v] = mp[v];
    out_pstar[eqPG] = out_pstar[eqRO] * a * a / g;
    return 0;
  }

  // ---------------------------------------------------------------------
  // HLLD / HLL (MHD): HLLD_MHD.cpp:124-333, 342-368, 377-417
  void HLLD_signal_speeds(const double *Pl, const double *Pr, const double g, double &Sl, double &Sr) const
  {
    double BX_ = 0.5 * (Pl[eqBX] + Pr[eqBX]);
    double cf_l = cfast_components(Pl[eqRO], Pl[eqPG], BX_, Pl[eqBY], Pl[eqBZ], g);
    double cf_r = cfast_components(Pr[eqRO], Pr[eqPG], BX_, Pr[eqBY], Pr[eqBZ], g);
    double cf_max = std::max(cf_l, cf_r);
    Sl = std::min(Pl[eqVX], Pr[eqVX]) - cf_max;
    Sr = std::max(Pl[eqVX], Pr[eqVX]) + cf_max;
  }
  int MHD_HLL_flux_solver(const double *Pl, const double *Pr, const double g, double *out_flux,
                          double *out_ustar) const
  {
    double UL[8], UR[8], FL[8], FR[8], lam0, lam1;
    mhd_PtoU(Pl, UL, g);
    mhd_PtoU(Pr, UR, g);
    mhd_PUtoFlux(Pl, UL, FL);
    mhd_PUtoFlux(Pr, UR, FR);
    HLLD_signal_speeds(Pl, Pr, g, lam0, lam1);
    if (lam0 > 0.0) {
      for (int v = 0; v < 8; v++) out_flux[v] = FL[v];
      for (int v = 0; v < 8; v++) out_ustar[v] = UL[v];
    }
    else if (lam1 < 0.0) {
      for (int v = 0; v < 8; v++) out_flux[v] = FR[v];
      for (int v = 0; v < 8; v++) out_ustar[v] = UR[v];
    }
    else {
      for (int v = 0; v < 8; v++)
        out_flux[v] = (lam1 * FL[v] - lam0 * FR[v] + lam1 * lam0 * (UR[v] - UL[v])) / (lam1 - lam0);
      for (int v = 0; v < 8; v++)
        out_ustar[v] = (lam1 * UR[v] - lam0 * UL[v] - FR[v] + FL[v]) / (lam1 - lam0);
    }
    return 0;
  }
  int MHD_HLLD_flux_solver(const double *Pl, const double *Pr, const double g, double *out_flux,
                           double *out_ustar) const
  {
    double UL[8], UR[8], FL[8], FR[8], ULs[8], URs[8], ULss[8], URss[8], lam[5];
    double BX_ = 0.5 * (Pl[eqBX] + Pr[eqBX]);
    mhd_PtoU(Pl, UL, g);
    mhd_PtoU(Pr, UR, g);
    mhd_PUtoFlux(Pl, UL, FL);
    mhd_PUtoFlux(Pr, UR, FR);
    HLLD_signal_speeds(Pl, Pr, g, lam[0], lam[4]);
    double sl_vl = lam[0] - Pl[eqVX];
    double sr_vr = lam[4] - Pr[eqVX];
    double tp_r = mhd_Ptot(Pr);
    double tp_l = mhd_Ptot(Pl);
    double temp = sr_vr * Pr[eqRO] - sl_vl * Pl[eqRO];
    lam[2] = (sr_vr * UR[eqMMX] - sl_vl * UL[eqMMX] - tp_r + tp_l) / temp;
    double tp_s = (sr_vr * Pr[eqRO] * tp_l - sl_vl * Pl[eqRO] * tp_r +
                   Pl[eqRO] * Pr[eqRO] * sr_vr * sl_vl * (Pr[eqVX] - Pl[eqVX])) /
                  temp;
    double sl_sm = lam[0] - lam[2];
    double sr_sm = lam[4] - lam[2];
    ULs[eqRHO] = Pl[eqRO] * sl_vl / sl_sm;
    URs[eqRHO] = Pr[eqRO] * sr_vr / sr_sm;
    ULs[eqMMX] = lam[2] * ULs[eqRHO];
    URs[eqMMX] = lam[2] * URs[eqRHO];
    double temp_l1 = lam[2] - Pl[eqVX];
    double temp_l2 = Pl[eqRO] * sl_vl * sl_sm - BX_ * BX_;
    double temp_r1 = lam[2] - Pr[eqVX];
    double temp_r2 = Pr[eqRO] * sr_vr * sr_sm - BX_ * BX_;
    double vys_l = Pl[eqVY], vys_r = Pr[eqVY], vzs_l = Pl[eqVZ], vzs_r = Pr[eqVZ];
    if (std::isfinite(temp_l1 / temp_l2)) {
      vys_l = Pl[eqVY] - BX_ * Pl[eqBY] * temp_l1 / temp_l2;
      vzs_l = Pl[eqVZ] - BX_ * Pl[eqBZ] * temp_l1 / temp_l2;
    }
    if (std::isfinite(temp_r1 / temp_r2)) {
      vys_r = Pr[eqVY] - BX_ * Pr[eqBY] * temp_r1 / temp_r2;
      vzs_r = Pr[eqVZ] - BX_ * Pr[eqBZ] * temp_r1 / temp_r2;
    }
    ULs[eqMMY] = vys_l * ULs[eqRHO];
    URs[eqMMY] = vys_r * URs[eqRHO];
    ULs[eqMMZ] = vzs_l * ULs[eqRHO];
    URs[eqMMZ] = vzs_r * URs[eqRHO];
    ULs[eqBBX] = URs[eqBBX] = BX_;
    temp_l1 = Pl[eqRO] * sl_vl * sl_vl - BX_ * BX_;
    temp_r1 = Pr[eqRO] * sr_vr * sr_vr - BX_ * BX_;
    ULs[eqBBY] = 0.0;
    URs[eqBBY] = 0.0;
    ULs[eqBBZ] = 0.0;
    URs[eqBBZ] = 0.0;
    if (std::isfinite(temp_l1 / temp_l2)) {
      ULs[eqBBY] = Pl[eqBY] * temp_l1 / temp_l2;
      ULs[eqBBZ] = Pl[eqBZ] * temp_l1 / temp_l2;
    }
    if (std::isfinite(temp_r1 / temp_r2)) {
      URs[eqBBY] = Pr[eqBY] * temp_r1 / temp_r2;
      URs[eqBBZ] = Pr[eqBZ] * temp_r1 / temp_r2;
    }
    temp_l1 = Pl[eqVX] * BX_ + Pl[eqVY] * Pl[eqBY] + Pl[eqVZ] * Pl[eqBZ];
    temp_r1 = Pr[eqVX] * BX_ + Pr[eqVY] * Pr[eqBY] + Pr[eqVZ] * Pr[eqBZ];
    temp_l2 = lam[2] * ULs[eqBBX] + vys_l * ULs[eqBBY] + vzs_l * ULs[eqBBZ];
    temp_r2 = lam[2] * URs[eqBBX] + vys_r * URs[eqBBY] + vzs_r * URs[eqBBZ];
    ULs[eqERG] = (sl_vl * UL[eqERG] - tp_l * Pl[eqVX] + tp_s * lam[2] + BX_ * (temp_l1 - temp_l2)) / sl_sm;
    URs[eqERG] = (sr_vr * UR[eqERG] - tp_r * Pr[eqVX] + tp_s * lam[2] + BX_ * (temp_r1 - temp_r2)) / sr_sm;
    lam[1] = lam[2] - std::fabs(BX_) / std::sqrt(ULs[eqRHO]);
    lam[3] = lam[2] + std::fabs(BX_) / std::sqrt(URs[eqRHO]);
    if (BX_ == 0) {
      for (int v = 0; v < 8; v++) {
        ULss[v] = ULs[v];
        URss[v] = URs[v];
      }
    }
    else {
      ULss[eqRHO] = ULs[eqRHO];
      URss[eqRHO] = URs[eqRHO];
      double sgn = (BX_ > 0) - (BX_ < 0);
      temp_l1 = std::sqrt(ULs[eqRHO]);
      temp_r1 = std::sqrt(URs[eqRHO]);
      temp = temp_l1 + temp_r1;
      ULss[eqMMX] = lam[2] * ULss[eqRHO];
      URss[eqMMX] = lam[2] * URss[eqRHO];
      double vy_ss = (temp_l1 * vys_l + temp_r1 * vys_r + (URs[eqBBY] - ULs[eqBBY]) * sgn) / temp;
      ULss[eqMMY] = vy_ss * ULss[eqRHO];
      URss[eqMMY] = vy_ss * URss[eqRHO];
      double vz_ss = (temp_l1 * vzs_l + temp_r1 * vzs_r + (URs[eqBBZ] - ULs[eqBBZ]) * sgn) / temp;
      ULss[eqMMZ] = vz_ss * ULss[eqRHO];
      URss[eqMMZ] = vz_ss * URss[eqRHO];
      ULss[eqBBX] = URss[eqBBX] = BX_;
      ULss[eqBBY] = URss[eqBBY] =
          (temp_l1 * URs[eqBBY] + temp_r1 * ULs[eqBBY] + temp_l1 * temp_r1 * (vys_r - vys_l) * sgn) / temp;
      ULss[eqBBZ] = URss[eqBBZ] =
          (temp_l1 * URs[eqBBZ] + temp_r1 * ULs[eqBBZ] + temp_l1 * temp_r1 * (vzs_r - vzs_l) * sgn) / temp;
      temp = lam[2] * ULss[eqBBX] + vy_ss * ULss[eqBBY] + vz_ss * ULss[eqBBZ];
      ULss[eqERG] = ULs[eqERG] - temp_l1 * (temp_l2 - temp) * sgn;
      URss[eqERG] = URs[eqERG] + temp_r1 * (temp_r2 - temp) * sgn;
    }
    if (lam[0] > 0) {
      for (int v = 0; v < 8; v++) out_flux[v] = FL[v];
      for (int v = 0; v < 8; v++) out_ustar[v] = UL[v];
    }
    else if (lam[1] >= 0) {
      for (int v = 0; v < 8; v++) out_flux[v] = FL[v] + lam[0] * (ULs[v] - UL[v]);
      for (int v = 0; v < 8; v++) out_ustar[v] = ULs[v];
    }
    else if (lam[2] >= 0) {
      for (int v = 0; v < 8; v++)
        out_flux[v] = FL[v] + lam[1] * ULss[v] - (lam[1] - lam[0]) * ULs[v] - lam[0] * UL[v];
      for (int v = 0; v < 8; v++) out_ustar[v] = ULss[v];
    }
    else if (lam[3] >= 0) {
      for (int v = 0; v < 8; v++)
        out_flux[v] = FR[v] + lam[3] * URss[v] - (lam[3] - lam[4]) * URs[v] - lam[4] * UR[v];
      for (int v = 0; v < 8; v++) out_ustar[v] = URss[v];
    }
    else if (lam[4] >= 0) {
      for (int v = 0; v < 8; v++) out_flux[v] = FR[v] + lam[4] * (URs[v] - UR[v]);
      for (int v = 0; v < 8; v++) out_ustar[v] = URs[v];
    }
    else {
      for (int v = 0; v < 8; v++) out_flux[v] = FR[v];
      for (int v = 0; v < 8; v++) out_ustar[v] = UR[v];
    }
    for (int v = 8; v < nvar; v++) out_flux[v] = 0.0;
    for (int v = 8; v < nvar; v++) out_ustar[v] = 0.0;
    return 0;
  }

  // ---------------------------------------------------------------------
  // inviscid_flux dispatch.  use_hll = outcome of the HLLD->HLL switch
  // (solver_eqn_mhd_adi.cpp:167-181), evaluated by the caller from the cells'
  // div v and grad p.
  int hydro_inviscid_flux(const double dx, const double *Pl, const double *Pr, double *flux,
                          double *pstar, const int solve_flag, const double g)
  {
    // solver_eqn_hydro_adi.cpp:94-201
    int err = 0;
    double ustar[MAXNV];
    for (int v = 0; v < nvar; v++) ustar[v] = 0.0;
    for (int v = 0; v < nvar; v++) flux[v] = 0.0;
    for (int v = 0; v < nvar; v++) pstar[v] = 0.0;
    gamma = g;
    if (solve_flag == FLUX_LF) {
      err += get_LaxFriedrichs_flux(Pl, Pr, flux, dx);
      for (int v = 0; v < nvar; v++) pstar[v] = 0.5 * (Pl[v] + Pr[v]);
    }
    else if (solve_flag == FLUX_FVS) {
      err += FVS_flux(Pl, Pr, flux, pstar);
    }
    else if (solve_flag == FLUX_RSlinear || solve_flag == FLUX_RSexact || solve_flag == FLUX_RShybrid) {
      err += JMs_riemann_solve(Pl, Pr, pstar, solve_flag, g);
      PtoFlux(pstar, flux, g);
    }
    else if (solve_flag == FLUX_RSroe) {
      err += Roe_flux_solver_symmetric(Pl, Pr, g, HC_etamax, pstar, flux);
    }
    else if (solve_flag == FLUX_RSroe_pv) {
      err += Roe_prim_var_solver(Pl, Pr, g, pstar);
      PtoFlux(pstar, flux, g);
    }
    else if (solve_flag == FLUX_RS_HLL) {
      err += hydro_HLL_flux_solver(Pl, Pr, g, flux, ustar);
      err += UtoP(ustar, pstar, MinTemperature, g);
    }
    else throw physics_error("what sort of flux solver do you mean???");
    return err;
  }
  int mhd_ideal_inviscid_flux(const double dx, const double *Pl, const double *Pr, double *flux,
                              double *pstar, const int solve_flag, const double g, const bool use_hll)
  {
    // solver_eqn_mhd_adi.cpp:102-200
    int err = 0;
    double ustar[MAXNV];
    for (int v = 0; v < nvar; v++) ustar[v] = 0.0;
    for (int v = 0; v < nvar; v++) flux[v] = 0.0;
    for (int v = 0; v < nvar; v++) pstar[v] = 0.0;
    if (solve_flag == FLUX_LF) {
      err += get_LaxFriedrichs_flux(Pl, Pr, flux, dx);
      for (int v = 0; v < nvar; v++) pstar[v] = 0.5 * (Pl[v] + Pr[v]);
    }
    else if (solve_flag == FLUX_RS_HLLD) {
      if (use_hll) err += MHD_HLL_flux_solver(Pl, Pr, g, flux, ustar);
      else err += MHD_HLLD_flux_solver(Pl, Pr, g, flux, ustar);
      err = UtoP(ustar, pstar, MinTemperature, g);
    }
    else if (solve_flag == FLUX_RS_HLL) {
      err += MHD_HLL_flux_solver(Pl, Pr, g, flux, ustar);
      err = UtoP(ustar, pstar, MinTemperature, g);
    }
    else if (solve_flag == FLUX_RSroe) {
      err += MHD_Roe_CV_flux_solver_symmetric(Pl, Pr, g, HC_etamax, pstar, flux);
      if (err) {  // (never: every step of the Roe solver returns 0)
        err = mhd_JMs_riemann_solve(Pl, Pr, pstar, 1, g);
        PtoFlux(pstar, flux, g);
      }
    }
    else if (solve_flag == FLUX_RSlinear || solve_flag == FLUX_RSexact || solve_flag == FLUX_RShybrid) {
      // modes 2 and 3 are fatal inside the solver (riemannMHD.cpp:176-181)
      err += mhd_JMs_riemann_solve(Pl, Pr, pstar, solve_flag, g);
      PtoFlux(pstar, flux, g);
    }
    else throw physics_error("what sort of flux solver do you mean???");
    return err;
  }
  int glm_inviscid_flux(const double dx, const double *Pl, const double *Pr, double *flux,
                        double *pstar, const int solve_flag, const double g, const bool use_hll)
  {
    // solver_eqn_mhd_adi.cpp:662-769
    for (int v = 0; v < nvar; v++) flux[v] = 0.0;
    for (int v = 0; v < nvar; v++) pstar[v] = 0.0;
    double left[MAXNV], right[MAXNV];
    for (int v = 0; v < nvar; v++) left[v] = Pl[v];
    for (int v = 0; v < nvar; v++) right[v] = Pr[v];
    double psistar = 0.5 * (left[eqSI] + right[eqSI] - (right[eqBX] - left[eqBX]));
    double bxstar = 0.5 * (left[eqBX] + right[eqBX] - (right[eqSI] - left[eqSI]));
    left[eqSI] = right[eqSI] = 0.0;
    left[eqBX] = right[eqBX] = bxstar;
    int err = mhd_ideal_inviscid_flux(dx, left, right, flux, pstar, solve_flag, g, use_hll);
    flux[eqERG] += GLM_chyp * bxstar * psistar;
    flux[eqBBX] = GLM_chyp * psistar;
    flux[eqPSI] = GLM_chyp * bxstar;
    return err;
  }
  int inviscid_flux(const double dx, const double *Pl, const double *Pr, double *flux, double *pstar,
                    const int solve_flag, const double g, const bool use_hll)
  {
    if (eqntype == EQEUL) return hydro_inviscid_flux(dx, Pl, Pr, flux, pstar, solve_flag, g);
    if (eqntype == EQMHD) return mhd_ideal_inviscid_flux(dx, Pl, Pr, flux, pstar, solve_flag, g, use_hll);
    return glm_inviscid_flux(dx, Pl, Pr, flux, pstar, solve_flag, g, use_hll);
  }

  // ---------------------------------------------------------------------
  // AVFalle: solver_eqn_hydro_adi.cpp:283-330 / solver_eqn_mhd_adi.cpp:209-286
  int AVFalle(const double *Pl, const double *Pr, const double *pstar, double *flux) const
  {
    if (eqntype == EQEUL) {
      double prefactor = chydro(pstar, gamma) * etav * pstar[eqRO];
      double momvisc = prefactor * (Pr[eqVX] - Pl[eqVX]);
      double ergvisc = momvisc * pstar[eqVX];
      flux[eqMMX] -= momvisc;
      momvisc = prefactor * (Pr[eqVY] - Pl[eqVY]);
      flux[eqMMY] -= momvisc;
      ergvisc += momvisc * pstar[eqVY];
      momvisc = prefactor * (Pr[eqVZ] - Pl[eqVZ]);
      flux[eqMMZ] -= momvisc;
      ergvisc += momvisc * pstar[eqVZ];
      flux[eqERG] -= ergvisc;
      return 0;
    }
    double prefactor = cfast_components(0.5 * (Pl[eqRO] + Pr[eqRO]), 0.5 * (Pl[eqPG] + Pr[eqPG]),
                                        0.5 * (Pl[eqBX] + Pr[eqBX]), 0.5 * (Pl[eqBY] + Pr[eqBY]),
                                        0.5 * (Pl[eqBZ] + Pr[eqBZ]), gamma) *
                       etav * pstar[eqRO];
    double momvisc = prefactor * (Pr[eqVX] - Pl[eqVX]);
    double ergvisc = momvisc * pstar[eqVX];
    flux[eqMMX] -= momvisc;
    momvisc = prefactor * (Pr[eqVY] - Pl[eqVY]);
    flux[eqMMY] -= momvisc;
    ergvisc += momvisc * pstar[eqVY];
    momvisc = prefactor * (Pr[eqVZ] - Pl[eqVZ]);
    flux[eqMMZ] -= momvisc;
    ergvisc += momvisc * pstar[eqVZ];
    prefactor *= etav / (etav * pstar[eqRO]);  // FV_etaB/(FV_etav*Pstar[eqRO])
    momvisc = prefactor * (Pr[eqBY] - Pl[eqBY]);
    flux[eqBBY] -= momvisc;
    ergvisc += momvisc * pstar[eqBY];
    momvisc = prefactor * (Pr[eqBZ] - Pl[eqBZ]);
    flux[eqBBZ] -= momvisc;
    ergvisc += momvisc * pstar[eqBZ];
    flux[eqERG] -= ergvisc;
    return 0;
  }

  // set_interface_tracer_flux: solver_eqn_base.cpp:281-342
  void set_interface_tracer_flux(const double *left, const double *right, double *flux) const
  {
    double corrector[MAXNV];
    for (int v = 0; v < nvar; v++) corrector[v] = 1.0;
    if (ntr > 0) {
      if (flux[eqRHO] > 0.0) {
        if (MP && MP->present) MP->sCMA(corrector, left);
        for (int t = 0; t < ntr; t++) flux[eqTR[t]] = left[eqTR[t]] * flux[eqRHO] * corrector[eqTR[t]];
      }
      else if (flux[eqRHO] < 0.0) {
        if (MP && MP->present) MP->sCMA(corrector, right);
        for (int t = 0; t < ntr; t++) flux[eqTR[t]] = right[eqTR[t]] * flux[eqRHO] * corrector[eqTR[t]];
      }
      else {
        for (int t = 0; t < ntr; t++) flux[eqTR[t]] = 0.0;
      }
    }
  }

  // InterCellFlux: solver_eqn_base.cpp:152-204.  hc_etamax is what
  // pre_calc_viscous_terms would have selected (only used for AV 3/4).
  int InterCellFlux(const double dx, const double *lp, const double *rp, double *f, double *pstar,
                    const int solver, const int artvisc, const double g, const double hc_etamax,
                    const bool use_hll)
  {
    gamma = g;
    if (artvisc == AV_HCORRECTION || artvisc == AV_HCORR_FKJ98) HC_etamax = hc_etamax;
    int err = inviscid_flux(dx, lp, rp, f, pstar, solver, g, use_hll);
    if (artvisc == AV_FKJ98_1D || artvisc == AV_HCORR_FKJ98) AVFalle(lp, rp, pstar, f);
    set_interface_tracer_flux(lp, rp, f);
    return err;
  }
};

}  // namespace orc
#endif
