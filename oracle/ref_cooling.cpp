// ref_cooling.cpp -- ORACLE side: the REFERENCE's own mp_only_cooling object under test.
//
// TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libpion_ref.so, see oracle/Makefile).
//
// microphysics/mp_only_cooling.cpp compiles here as it lies (no GSL in it); what does not are the
// sources of three of its base classes -- cooling_function_SD93CIE (cooling_SD93_cie.cpp),
// Hummer94_Hrecomb (hydrogen_recomb_Hummer94.cpp) and CoolingFn (cooling.cpp) -- which evaluate
// their rate curves with GSL splines (tools/interpolate.h).  Those three classes get TEST DOUBLES
// below: their member functions, declared by the reference's own unmodified headers, return a rate
// curve SUPPLIED BY THE TEST through ref_cooling_set_rate_curves() (three double(double) function
// pointers), exactly as HarnessODE in ref_harness.cpp feeds a test-supplied rate into the
// reference's Integrator_Base.  They are not a GSL stand-in: no GSL header, symbol or algorithm is
// imitated, and the curve VALUES stay "parity unpinned" (the tests hand over the product's natural
// cubic splines, themselves cross-checked against scipy).  Everything downstream of the three
// curves is the reference's compiled code:
//   gen_mpoc_lookup_tables (mp_only_cooling.cpp:528-579)   table grid, 5 tables, slopes
//   Edot_WSS09CIE_heat_cool_metallines (:491-521)          bisection + interpolation + formula
//   TimeUpdateMP (:167-218), dPdt (:227-236)               clamps, E<->T mapping, Cash-Karp driver
//   timescales (:333-368)                                  cooling time
//   Set_Temp / Temperature (:245-280)
//
// Compiled with -fno-access-control (this file only) so that the fixtures can read the private
// table `lt` and call the private Edot().
#include "tools/reporting.h"
#include "tools/mem_manage.h"
#include "constants.h"
#include "sim_params.h"
#include "microphysics/mp_only_cooling.h"

#include <cmath>
#include <string>

// ------------------------------------------------------------------ test-supplied rate curves
typedef double (*rate_fn)(double);
static rate_fn g_cie = 0, g_rrr = 0, g_tot = 0;

// ---- test double: cooling_function_SD93CIE (declared in microphysics/cooling_SD93_cie.h)
cooling_function_SD93CIE::cooling_function_SD93CIE()
    : Nspl(0), Tarray(0), Larray(0), spline_id(-1), MaxTemp(0), MinTemp(0), MaxSlope(0), MinSlope(0),
      have_set_cooling(false)
{
}
cooling_function_SD93CIE::~cooling_function_SD93CIE() {}
void cooling_function_SD93CIE::setup_SD93_cie() { have_set_cooling = true; }
void cooling_function_SD93CIE::setup_WSS09_CIE() { have_set_cooling = true; }
void cooling_function_SD93CIE::setup_WSS09_CIE_OnlyMetals() { have_set_cooling = true; }
double cooling_function_SD93CIE::cooling_rate_SD93CIE(const double T)
{
  if (!g_cie) rep.error("test double: no CIE rate curve supplied", 0);
  return g_cie(T);
}
// ---- test double: Hummer94_Hrecomb (microphysics/hydrogen_recomb_Hummer94.h)
Hummer94_Hrecomb::Hummer94_Hrecomb()
    : kB(1.381e-16), hr_Nspl(0), hr_t(0), hr_alpha(0), hr_beta(0), hr_btot(0)
{
}
Hummer94_Hrecomb::~Hummer94_Hrecomb() {}
double Hummer94_Hrecomb::Hii_rad_recomb_rate(double T)
{
  if (!g_rrr) rep.error("test double: no recombination rate curve supplied", 0);
  return g_rrr(T);
}
double Hummer94_Hrecomb::Hii_total_cooling(double T)
{
  if (!g_tot) rep.error("test double: no H+ cooling rate curve supplied", 0);
  return g_tot(T);
}
// ---- test double: CoolingFn (microphysics/cooling.h); never evaluated for EP.cooling = 8
CoolingFn::CoolingFn(int f) : WhichFunction(f), Temp(0), Lamb(0), Lam2(0), Nspl(0) {}
CoolingFn::~CoolingFn() {}
double CoolingFn::CoolingRate(const double, const double, const double, const double, const double)
{
  rep.error("test double: CoolingFn::CoolingRate is not on the tested path", 0);
  return 0.0;
}

// ------------------------------------------------------------------ the object under test
struct RefCooling {
  which_physics EP;
  rad_sources RS;
  std::string trnames[16];
  mp_only_cooling *mp;
  double gamma;
  int nvar;
};

extern "C" {

int ref_cooling_have_curves() { return g_cie && g_rrr && g_tot; }

void ref_cooling_set_rate_curves(void *cie, void *rrr, void *tot)
{
  g_cie = (rate_fn)cie;
  g_rrr = (rate_fn)rrr;
  g_tot = (rate_fn)tot;
}

// for RefSim (ref_harness.cpp): the reference's mp_only_cooling as the global MP object
microphysics_base *ref_cooling_new_mp(int nv, int ntr, const std::string *tr, which_physics *ep,
                                      rad_sources *rs)
{
  return new mp_only_cooling(nv, ntr, tr, ep, rs);
}

int ref_cooling_create(double min_temp, double max_temp, double gamma, int nvar, int ntracer, void **h)
{
  if (!ref_cooling_have_curves()) return -1;
  RefCooling *c = new RefCooling;
  c->EP.dynamics = 1;
  c->EP.raytracing = 0;
  c->EP.cooling = 8;
  c->EP.chemistry = 0;
  c->EP.coll_ionisation = 0;
  c->EP.rad_recombination = 0;
  c->EP.phot_ionisation = 0;
  c->EP.update_erg = 1;
  c->EP.MP_timestep_limit = 1;
  c->EP.MinTemperature = min_temp;
  c->EP.MaxTemperature = max_temp;
  c->RS.Nsources = 0;
  for (int t = 0; t < 16; t++) c->trnames[t] = "colour";
  c->gamma = gamma;
  c->nvar = nvar;
  c->mp = new mp_only_cooling(nvar, ntracer, c->trnames, &c->EP, &c->RS);
  *h = c;
  return 0;
}
void ref_cooling_destroy(void *h)
{
  RefCooling *c = (RefCooling *)h;
  delete c->mp;
  delete c;
}
// gen_mpoc_lookup_tables' product: T[NT], tabs[5][NT] = {rrhp, C_rrh, C_ffhe, C_fbdn, C_cie}, slopes[5][NT]
int ref_cooling_tables(void *h, int nT, double *T, double *tabs, double *slopes)
{
  mp_only_cooling *m = ((RefCooling *)h)->mp;
  if ((size_t)nT != m->lt.NT) return (int)m->lt.NT;
  const std::vector<double> *t[5] = {&m->lt.rrhp, &m->lt.C_rrh, &m->lt.C_ffhe, &m->lt.C_fbdn, &m->lt.C_cie};
  const std::vector<double> *s[5] = {&m->lt.s_rrhp, &m->lt.s_C_rrh, &m->lt.s_C_ffhe, &m->lt.s_C_fbdn,
                                     &m->lt.s_C_cie};
  for (int i = 0; i < nT; i++) {
    T[i] = m->lt.T[i];
    for (int k = 0; k < 5; k++) {
      tabs[k * nT + i] = (*t[k])[i];
      slopes[k * nT + i] = (*s[k])[i];
    }
  }
  return 0;
}
// {MinT_allowed, MaxT_allowed, Mu_tot_over_kB} as the constructor left them (mp_only_cooling.cpp:78-139)
void ref_cooling_limits(void *h, double *out)
{
  mp_only_cooling *m = ((RefCooling *)h)->mp;
  out[0] = m->MinT_allowed;
  out[1] = m->MaxT_allowed;
  out[2] = m->Mu_tot_over_kB;
}
int ref_cooling_edot(void *h, int n, const double *rho, const double *T, double *out)
{
  mp_only_cooling *m = ((RefCooling *)h)->mp;
  for (int i = 0; i < n; i++) out[i] = m->Edot(rho[i], T[i]);
  return 0;
}
// TimeUpdateMP on n independent states of nvar primitives; Tf[n] = the temperature it reports
int ref_cooling_update(void *h, int n, double dt, const double *Pin, double *Pout, double *Tf)
{
  RefCooling *c = (RefCooling *)h;
  int err = 0;
  for (int i = 0; i < n; i++)
    err += c->mp->TimeUpdateMP(Pin + (size_t)i * c->nvar, Pout + (size_t)i * c->nvar, dt, c->gamma, 0, Tf + i);
  return err;
}
int ref_cooling_timescale(void *h, int n, const double *Pin, double *out)
{
  RefCooling *c = (RefCooling *)h;
  for (int i = 0; i < n; i++) out[i] = c->mp->timescales(Pin + (size_t)i * c->nvar, c->gamma, true, false, false);
  return 0;
}
}  // extern "C"
