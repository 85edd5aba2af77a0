// dev_cooling.h -- per-cell radiative cooling/heating on the device.
//
// mp_only_cooling with EP.cooling = 8 (WSS09_CIE_LINE_HEAT_COOL):
//   Edot            microphysics/mp_only_cooling.cpp:491-521 (bisection in a 200-point
//                   log-T table + linear interpolation of 5 rate tables)
//   TimeUpdateMP    microphysics/mp_only_cooling.cpp:167-218
//   timescales      microphysics/mp_only_cooling.cpp:333-368
//   Cash-Karp RK5   microphysics/integrator.cpp:42-84 (tableau), 285-392 (Step_RK5CK incl. the
//                   first-order shortcut), 401-531 (Stepper_RKCK, BISECTION_STEPPER),
//                   540-602 (Int_Adaptive_RKCK)
// No transcendental function is evaluated per cell: the tables are built once on
// the host and handed over through pion_gpu_set_cooling_tables.
#ifndef PION_DEV_COOLING_H
#define PION_DEV_COOLING_H

#include "kernels.h"

namespace pion {

struct Cooling {
  static PDEV double edot(const CoolDev &c, const double rho, const double T)
  {
    const int NT = c.NT;
    // The table interval: the reference bisects (mp_only_cooling.cpp:496-503), which ends on the largest i in
    // [0, NT-2] with T[i] < T (0 if there is none; NaN -> 0).  On the log-spaced grid the same i comes from a
    // single-precision logarithm as a first guess, corrected by stepping along the table itself until exactly that
    // condition holds: the same index for every T, bit-identical results, an eighth of the dependent look-ups.
    int ilo = 0;
    if (c.inv_dlg > 0.0f) {
      float gi = (T > 0.0) ? (__log2f((float)T) - c.lg0) * c.inv_dlg : 0.0f;
      gi = fminf(fmaxf(gi, 0.0f), (float)(NT - 2));
      ilo = (int)gi;
      while (ilo > 0 && !(c.T[ilo] < T)) ilo--;
      while (ilo < NT - 2 && c.T[ilo + 1] < T) ilo++;
    }
    else {
      int ihi = NT - 1, imid = 0;
      do {
        imid = ilo + (int)floor((ihi - ilo) / 2.0);
        if (c.T[imid] < T) ilo = imid;
        else ihi = imid;
      } while (ihi - ilo > 1);
    }
    const int iT = ilo;
    const double dT = T - c.T[iT];
    const double rho2 = rho * rho;
    double rate = 0.0;
    // tab rows: 0 rrhp, 1 C_rrh, 2 C_ffhe, 3 C_fbdn, 4 C_cie
    rate = -(c.tab[3 * NT + iT] + dT * c.slope[3 * NT + iT]) * rho2 * c.inv_Mu2_elec_H;
    rate = dmin(rate, -(c.tab[4 * NT + iT] + dT * c.slope[4 * NT + iT]) * rho2 * c.inv_Mu2);
    rate -= (c.tab[1 * NT + iT] + dT * c.slope[1 * NT + iT]) * rho2 * c.inv_Mu2_elec_H;
    rate -= (c.tab[2 * NT + iT] + dT * c.slope[2 * NT + iT]) * rho2 * c.inv_Mu2_elec_H;
    rate += 8.01e-12 * (c.tab[0 * NT + iT] + dT * c.slope[0 * NT + iT]) * rho2 * c.inv_Mu2_elec_H;
    return rate;
  }
  // mp_only_cooling::dPdt
  static PDEV double dPdt(const CoolDev &c, const double rho, const double gamma, const double E)
  {
    return edot(c, rho, E * (gamma - 1.0) * c.Mu_tot_over_kB / rho);
  }
  static PDEV void step_rk5ck(const CoolDev &c, const double rho, const double gamma, const double p0,
                              const double dt, double *pf, double *dp)
  {
    const double b21 = 0.2, b31 = 3. / 40., b32 = 9. / 40., b41 = 0.3, b42 = -0.9, b43 = 1.2, b51 = -11. / 54.,
                 b52 = 2.5, b53 = -70. / 27., b54 = 35. / 27., b61 = 1631. / 55296., b62 = 175. / 512.,
                 b63 = 575. / 13824., b64 = 44275. / 110592., b65 = 253. / 4096., c1 = 37. / 378.,
                 c3 = 250. / 621., c4 = 125. / 594., c6 = 512. / 1771.;
    const double dc1 = c1 - 2825. / 27648., dc3 = c3 - 18575. / 48384., dc4 = c4 - 13525. / 55296.,
                 dc5 = -277. / 14336., dc6 = c6 - 0.25;
    double k1, k2, k3, k4, k5, k6, ptemp;
    k1 = dPdt(c, rho, gamma, p0);
    ptemp = 0.0;
    ptemp += fabs(k1) * dt / (p0 + 1.0e-100);
    if (ptemp < 1.e-6) {
      *pf = p0 + k1 * dt;
      *dp = k1 * dt;
      return;
    }
    k1 *= dt;
    ptemp = p0 + b21 * k1;
    k2 = dPdt(c, rho, gamma, ptemp);
    k2 *= dt;
    ptemp = p0 + b31 * k1 + b32 * k2;
    k3 = dPdt(c, rho, gamma, ptemp);
    k3 *= dt;
    ptemp = p0 + b41 * k1 + b42 * k2 + b43 * k3;
    k4 = dPdt(c, rho, gamma, ptemp);
    k4 *= dt;
    ptemp = p0 + b51 * k1 + b52 * k2 + b53 * k3 + b54 * k4;
    k5 = dPdt(c, rho, gamma, ptemp);
    k5 *= dt;
    ptemp = p0 + b61 * k1 + b62 * k2 + b63 * k3 + b64 * k4 + b65 * k5;
    k6 = dPdt(c, rho, gamma, ptemp);
    k6 *= dt;
    *pf = p0 + c1 * k1 + c3 * k3 + c4 * k4 + c6 * k6;
    *dp = dc1 * k1 + dc3 * k3 + dc4 * k4 + dc5 * k5 + dc6 * k6;
  }
  static PDEV int stepper_rkck(const CoolDev &c, const double rho, const double gamma, const double p0,
                               const double t0, const double htry, const double errtol, double *p1,
                               double *hdid, double *hnext)
  {
    int rval = 0;
    double h = htry;
    if (h < 0) return 1;
    double tnew = t0;
    const double eps = 1.e-100;
    double maxerr, err = 0.0, ptemp = 0.0;
    int ct = 0;
    do {
      step_rk5ck(c, rho, gamma, p0, h, &ptemp, &err);
      maxerr = 0;
      if (!isfinite(err) || !isfinite(ptemp) || ptemp < 0.0) {
        maxerr = dmax(maxerr, 1000.0);
      }
      else {
        err /= fabs(ptemp) + eps;
        err = fabs(err / errtol);
        maxerr = dmax(maxerr, err);
      }
      if (maxerr > 1.) h /= 2.0;
      tnew = t0 + h;
      if (tnew == t0) return -2;
      ct++;
    } while (maxerr > 1.0 && ct < 50);
    if (maxerr > 1.0) rval += ct + static_cast<int>(fabs(maxerr));
    *hnext = h * 2.0;
    *hdid = h;
    *p1 = ptemp;
    if (isnan(*p1) || isinf(*p1)) {
      *p1 = -1.e100;
      rval++;
    }
    return rval;
  }
  static PDEV int int_adaptive_rkck(const CoolDev &c, const double rho, const double gamma, const double p0,
                                    const double t0, const double dt, const double errtol, double *pf)
  {
    double t = t0, p1 = p0, p2 = 0.0;
    const double tf = t0 + dt;
    double h = dt, hdid = 0.0, hnext = 0.0;
    int err = 0, ct = 0;
    const int ctmax = 25;
    do {
      err += stepper_rkck(c, rho, gamma, p1, t, h, errtol, &p2, &hdid, &hnext);
      t += hdid;
      h = dmin(hnext, tf - t);
      ct++;
      p1 = p2;
    } while (t < tf && (err == 0) && (ct < ctmax));
    *pf = p1;
    return err;
  }
  // TimeUpdateMP: returns the new pressure (all other primitives are unchanged)
  static PDEV double time_update(const CoolDev &c, const double rho, const double pg_in, const double dt,
                                 const double g, int &errbits)
  {
    if (!isfinite(rho) || !isfinite(pg_in)) errbits |= ERR_COOLING;
    const double Eint0 = pg_in / (g - 1.0);
    double Eint = Eint0;
    int e = int_adaptive_rkck(c, rho, g, Eint, 0.0, dt, 1.0e-2, &Eint);
    if (e) errbits |= ERR_COOLING;
    double pg = Eint * (g - 1);
    double Tf = pg * c.Mu_tot_over_kB / rho;
    if (Tf > c.MaxT_allowed) pg *= c.MaxT_allowed / (Tf);
    else if (Tf < c.MinT_allowed) pg *= c.MinT_allowed / (Tf);
    return pg;
  }
  static PDEV double timescale(const CoolDev &c, const double rho, const double pg, const double gam)
  {
    double Eint = pg / (gam - 1.0);
    double T = pg * c.Mu_tot_over_kB / rho;
    double mintime = 1.0e99;
    if (T >= 1.1 * c.MinT_allowed) {
      double rate = dmax(fabs(edot(c, rho, T)), fabs(edot(c, rho, dmax(c.MinT_allowed, 0.5 * T))));
      mintime = dmin(mintime, Eint / rate);
    }
    return mintime;
  }
};

}  // namespace pion
#endif
