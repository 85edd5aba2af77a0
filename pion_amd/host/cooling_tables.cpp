// cooling_tables.cpp -- host-side builder of the mp_only_cooling look-up tables.
//
// Restates, for EP.cooling = 8 (WSS09_CIE_LINE_HEAT_COOL), what the reference does once at
// start-up: mp_only_cooling::gen_mpoc_lookup_tables (microphysics/mp_only_cooling.cpp:528-579)
// on top of cooling_function_SD93CIE::cooling_rate_SD93CIE (cooling_SD93_cie.cpp:666-704, natural
// cubic spline through the WSS09 metals-only curve in log-log space, power-law extrapolation)
// and Hummer94_Hrecomb::{Hii_rad_recomb_rate, Hii_total_cooling}
// (hydrogen_recomb_Hummer94.cpp:38-147,165-278, natural cubic splines of rate/sqrt(T) in T).
//
// The reference evaluates its splines with GSL's gsl_interp_cspline (natural boundary
// conditions; tools/interpolate.cpp:73-110).  GSL is not available here, so the spline is
// restated from its published definition and this boundary is "parity unpinned": it is
// cross-checked against scipy's natural CubicSpline (tests/test_cooling.py), not against the
// reference.
//
// C-ABI (plain pointers), called by the host driver before pion_gpu_set_cooling_tables.
#include <cmath>
#include <vector>

#include "cooling_data.h"

namespace {

// natural cubic spline in the representation GSL uses (cspline.c): second-derivative
// coefficients c[i] with c[0]=c[n-1]=0; eval: y_i + d*(b_i + d*(c_i + d*d_i))
struct Spline {
  std::vector<double> x, y, c;
  void init(const double *xa, const double *ya, int n)
  {
    x.assign(xa, xa + n);
    y.assign(ya, ya + n);
    c.assign(n, 0.0);
    const int sys = n - 2;  // unknowns c[1..n-2]
    if (sys <= 0) return;
    std::vector<double> diag(sys), off(sys), g(sys);
    for (int i = 0; i < sys; i++) {
      const double h_i = x[i + 1] - x[i];
      const double h_ip1 = x[i + 2] - x[i + 1];
      const double ydiff_i = y[i + 1] - y[i];
      const double ydiff_ip1 = y[i + 2] - y[i + 1];
      const double g_i = (h_i != 0.0) ? 1.0 / h_i : 0.0;
      const double g_ip1 = (h_ip1 != 0.0) ? 1.0 / h_ip1 : 0.0;
      off[i] = h_ip1;
      diag[i] = 2.0 * (h_ip1 + h_i);
      g[i] = 3.0 * (ydiff_ip1 * g_ip1 - ydiff_i * g_i);
    }
    // symmetric tridiagonal solve (Cholesky-like LDL^T, as gsl_linalg_solve_symm_tridiag)
    std::vector<double> gamma(sys), alpha(sys), cc(sys), z(sys);
    alpha[0] = diag[0];
    gamma[0] = (sys > 1) ? off[0] / alpha[0] : 0.0;
    for (int i = 1; i < sys - 1; i++) {
      alpha[i] = diag[i] - off[i - 1] * gamma[i - 1];
      gamma[i] = off[i] / alpha[i];
    }
    if (sys > 1) alpha[sys - 1] = diag[sys - 1] - off[sys - 2] * gamma[sys - 2];
    z[0] = g[0];
    for (int i = 1; i < sys; i++) z[i] = g[i] - gamma[i - 1] * z[i - 1];
    for (int i = 0; i < sys; i++) cc[i] = z[i] / alpha[i];
    std::vector<double> sol(sys);
    sol[sys - 1] = cc[sys - 1];
    for (int i = sys - 2; i >= 0; i--) sol[i] = cc[i] - gamma[i] * sol[i + 1];
    for (int i = 0; i < sys; i++) c[i + 1] = sol[i];
  }
  double eval(double xv) const
  {
    const int n = (int)x.size();
    int lo = 0, hi = n - 1;
    while (hi > lo + 1) {  // gsl_interp_bsearch
      const int i = (hi + lo) / 2;
      if (x[i] > xv) hi = i;
      else lo = i;
    }
    const int i = lo;
    const double dx = x[i + 1] - x[i];
    const double dy = y[i + 1] - y[i];
    const double delx = xv - x[i];
    const double b_i = (dy / dx) - dx * (c[i + 1] + 2.0 * c[i]) / 3.0;
    const double d_i = (c[i + 1] - c[i]) / (3.0 * dx);
    return y[i] + delx * (b_i + delx * (c[i] + delx * d_i));
  }
};

struct Tables {
  Spline wss, ha, hb, ht;
  double wssMinT, wssMaxT, wssMinSlope, wssMaxSlope;
  double hT[31], halpha[31], hbeta[31], hbtot[31];
  double hMinT, hMaxT, minA, maxA, minT_, maxT_;
  Tables()
  {
    wss.init(WSS09_logT, WSS09_logL, 91);
    wssMinT = WSS09_logT[0];
    wssMaxT = WSS09_logT[90];
    wssMinSlope = 8.0;  // cooling_SD93_cie.cpp:530
    wssMaxSlope = (WSS09_logL[90] - WSS09_logL[89]) / (WSS09_logT[90] - WSS09_logT[89]);
    for (int i = 0; i < 31; i++) {
      hT[i] = std::exp(std::log(10.0) * (1.0 + 0.2 * static_cast<double>(i)));
      halpha[i] = H94_caseB[i] / std::sqrt(hT[i]);
      hbeta[i] = H94_coolB[i] / std::sqrt(hT[i]);
      hbtot[i] = H94_coolTot[i] / std::sqrt(hT[i]);
    }
    ha.init(hT, halpha, 31);
    ht.init(hT, hbtot, 31);
    hMinT = hT[0];
    hMaxT = hT[30];
    minA = (std::log10(halpha[1]) - std::log10(halpha[0])) / (std::log10(hT[1]) - std::log10(hT[0]));
    maxA = (std::log10(halpha[30]) - std::log10(halpha[29])) / (std::log10(hT[30]) - std::log10(hT[29]));
    minT_ = (std::log10(hbtot[1]) - std::log10(hbtot[0])) / (std::log10(hT[1]) - std::log10(hT[0]));
    maxT_ = (std::log10(hbtot[30]) - std::log10(hbtot[29])) / (std::log10(hT[30]) - std::log10(hT[29]));
  }
  // cooling_function_SD93CIE::cooling_rate_SD93CIE
  double cooling_rate(double T) const
  {
    if (T < 0.0 || !std::isfinite(T)) return HUGE_VAL;
    double rate;
    T = std::log10(T);
    if (T > wssMaxT) rate = WSS09_logL[90] + wssMaxSlope * (T - wssMaxT);
    else if (T < wssMinT) rate = WSS09_logL[0] + wssMinSlope * (T - wssMinT);
    else rate = wss.eval(T);
    return std::exp(2.3025850929940459 * rate);  // pconst.ln10(), constants.h:44
  }
  double rrr(double T) const
  {
    if (T < 0.0 || !std::isfinite(T)) return HUGE_VAL;
    if (T > hMaxT) return halpha[30] * std::pow(T / hMaxT, maxA);
    if (T < hMinT) return halpha[0] * std::pow(T / hMinT, minA);
    return ha.eval(T);
  }
  double total_cooling(double T) const
  {
    const double kB = 1.381e-16;  // Hummer94_Hrecomb::kB
    if (T < 0.0 || !std::isfinite(T)) return HUGE_VAL;
    double rate;
    if (T > hMaxT) rate = hbtot[30] * std::pow(T / hMaxT, maxT_);
    else if (T < hMinT) rate = hbtot[0] * std::pow(T / hMinT, minT_);
    else rate = ht.eval(T);
    return rate * kB * T;
  }
};

}  // namespace

extern "C" {

// mp_only_cooling::gen_mpoc_lookup_tables: T[nT], tabs[5][nT] = {rrhp, C_rrh, C_ffhe, C_fbdn, C_cie},
// slopes[5][nT].  Returns 0.
int pion_host_build_cooling_tables(double min_temp, double max_temp, int nT, double *T, double *tabs, double *slopes)
{
  if (nT < 2 || !(min_temp > 0.0) || !(max_temp > min_temp)) return -1;
  static const Tables tb;
  const double dlogT = (std::log10(max_temp) - std::log10(min_temp)) / (nT - 1);
  for (int i = 0; i < nT; i++) T[i] = std::pow(10.0, std::log10(min_temp) + i * dlogT);
  double *rrhp = tabs, *C_rrh = tabs + nT, *C_ffhe = tabs + 2 * nT, *C_fbdn = tabs + 3 * nT, *C_cie = tabs + 4 * nT;
  for (int i = 0; i < nT; i++) {
    rrhp[i] = tb.rrr(T[i]);
    C_rrh[i] = tb.total_cooling(T[i]);
    C_ffhe[i] = 6.72e-28 * std::sqrt(T[i]);
    C_fbdn[i] = 1.20e-22 * std::exp(-33610.0 / T[i] - (2180.0 * 2180.0 / T[i] / T[i])) * std::exp(-T[i] * T[i] / 5.0e10);
    C_cie[i] = tb.cooling_rate(T[i]);
  }
  for (int k = 0; k < 5; k++) {
    const double *t = tabs + k * nT;
    double *s = slopes + k * nT;
    for (int i = 0; i < nT - 1; i++) s[i] = (t[i + 1] - t[i]) / (T[i + 1] - T[i]);
    s[nT - 1] = 0.0;
  }
  return 0;
}

double pion_host_cooling_rate_wss09(double T)
{
  static const Tables tb;
  return tb.cooling_rate(T);
}
double pion_host_hii_rrr(double T)
{
  static const Tables tb;
  return tb.rrr(T);
}
double pion_host_hii_total_cooling(double T)
{
  static const Tables tb;
  return tb.total_cooling(T);
}
}
