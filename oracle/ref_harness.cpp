// ref_harness.cpp -- ORACLE side: drives the REFERENCE's own solver objects.
//
// TEST INFRASTRUCTURE ONLY.  This file is compiled (oracle/Makefile, target
// _ref) together with the reference sources *where they lie* under
// /root/reference/source -- equations/, Riemann_solvers/, coord_sys/,
// spatial_solvers/, grid/cell_interface, the uniform-grid boundary updaters
// of boundaries/, microphysics/{microphysics_base,integrator} and the tools
// they need -- into oracle/_ref/libpion_ref.so.  No reference source is copied
// into this repository and no header/library stand-in is written: the parts of
// the reference that need GSL / SUNDIALS / Silo (UniformGrid, time_integrator,
// calc_timestep, the cooling tables) are NOT built.  What this harness adds is
//   * a GridBaseClass implementation over cells made by the reference's own
//     cell_interface (ids/positions/neighbours as UniformGrid makes them,
//     grid/uniform_grid.cpp:441-811), and the boundary lists of SetupBCs
//     (:1009-1216),
//   * the loops of time_integrator (sim_control/time_integrator.cpp:151-958)
//     and calc_dynamics_dt (sim_control/calc_timestep.cpp:271-333), written
//     against the reference's FV_solver_base interface,
// so that every flux, slope, source term, cell update, time-step and boundary
// fill below is computed by the reference's code.
//
// One live handle per process (the reference keeps global singletons CI, MP).

#include "tools/reporting.h"
#include "tools/mem_manage.h"
#include "constants.h"
#include "sim_params.h"
#include "grid/cell_interface.h"
#include "grid/grid_base_class.h"
#include "spatial_solvers/solver_eqn_base.h"
#include "spatial_solvers/solver_eqn_hydro_adi.h"
#include "spatial_solvers/solver_eqn_mhd_adi.h"
#include "microphysics/microphysics_base.h"
#include "microphysics/integrator.h"
#include "boundaries/boundaries.h"
#include "boundaries/periodic_boundaries.h"
#include "boundaries/outflow_boundaries.h"
#include "boundaries/oneway_out_boundaries.h"
#include "boundaries/inflow_boundaries.h"
#include "boundaries/reflecting_boundaries.h"
#include "boundaries/fixed_boundaries.h"
#include "boundaries/double_Mach_ref_boundaries.h"
#include "boundaries/jet_boundaries.h"
#include "boundaries/axisymmetric_boundaries.h"
#include "boundaries/jetreflect_boundaries.h"

#include "../include/pion_gpu.h"

#include <cstring>
#include <vector>

using namespace std;

// ---------------------------------------------------------------------------
// Grid over reference cells.
// ---------------------------------------------------------------------------
class HarnessGrid : public GridBaseClass {
 public:
  int nd, nv;
  int ng[3], nbc[3], nga[3];
  double dx, xmin[3], xmax[3];
  long ncell;
  std::vector<cell *> cells;
  cell *first_gd, *last_gd;

  long id(int ix, int iy, int iz) const
  {
    return (long)(ix + nbc[0]) + (long)nga[0] * ((iy + nbc[1]) + (long)nga[1] * (iz + nbc[2]));
  }
  HarnessGrid(const pion_gpu_config &c)
  {
    nd = c.ndim;
    nv = c.nvar;
    dx = c.dx;
    ncell = 1;
    for (int a = 0; a < 3; a++) {
      ng[a] = (a < nd) ? c.ng[a] : 1;
      nbc[a] = (a < nd) ? c.nbc : 0;
      nga[a] = ng[a] + 2 * nbc[a];
      ncell *= nga[a];
      xmin[a] = c.xmin[a];
      xmax[a] = c.xmin[a] + ng[a] * c.dx;
    }
    Wind = 0;
    RT = 0;
    cells.resize(ncell);
    for (long i = 0; i < ncell; i++) cells[i] = CI.new_cell();
    first_gd = last_gd = 0;
    for (int iz = -nbc[2]; iz < ng[2] + nbc[2]; iz++)
      for (int iy = -nbc[1]; iy < ng[1] + nbc[1]; iy++)
        for (int ix = -nbc[0]; ix < ng[0] + nbc[0]; ix++) {
          long k = id(ix, iy, iz);
          cell *cc = cells[k];
          cc->id = k;
          int ii[3] = {ix, iy, iz};
          double dpos[3];
          for (int a = 0; a < nd; a++) dpos[a] = xmin[a] + dx * (ii[a] + 0.5);
          CI.set_pos(cc, dpos);
          bool on = true;
          for (int a = 0; a < nd; a++)
            if (ii[a] < 0 || ii[a] >= ng[a]) on = false;
          cc->isgd = on;
          cc->isbd = !on;
          cc->isdomain = on;
          cc->isleaf = true;
          cc->timestep = true;
          cc->rt = false;
          // isedge for ghost cells: uniform_grid.cpp:581-606
          cc->isedge = 0;
          for (int a = 0; a < nd; a++) {
            if (ii[a] < 0) cc->isedge = ii[a];
            else if (ii[a] >= ng[a]) cc->isedge = ng[a] - 1 - ii[a];
          }
          for (int d = 0; d < 6; d++) cc->ngb[d] = 0;
          for (int a = 0; a < nd; a++) {
            int jj[3] = {ix, iy, iz};
            jj[a] = ii[a] - 1;
            if (jj[a] >= -nbc[a]) cc->ngb[2 * a] = cells[id(jj[0], jj[1], jj[2])];
            jj[a] = ii[a] + 1;
            if (jj[a] < ng[a] + nbc[a]) cc->ngb[2 * a + 1] = cells[id(jj[0], jj[1], jj[2])];
          }
          cc->npt_all = (k + 1 < ncell) ? cells[k + 1] : 0;
          cc->npt = 0;
          if (on) {
            if (!first_gd) first_gd = cc;
            if (last_gd) last_gd->npt = cc;
            last_gd = cc;
          }
        }
  }
  ~HarnessGrid()
  {
    for (size_t i = 0; i < BC_bd.size(); i++) {
      if (BC_bd[i]->refval) BC_bd[i]->refval = mem.myfree(BC_bd[i]->refval);
      delete BC_bd[i];
    }
    for (long i = 0; i < ncell; i++) CI.delete_cell(cells[i]);
  }

  cell *FirstPt() { return first_gd; }
  cell *FirstPt_All() { return cells[0]; }
  cell *LastPt() { return last_gd; }
  cell *LastPt_All() { return cells[ncell - 1]; }
  cell *NextPt(const cell *c, const enum direction d) { return c->ngb[d]; }
  cell *NextPt(const cell *c) { return c->npt; }
  cell *NextPt_All(const cell *c) { return c->npt_all; }
  class cell *PrevPt(const class cell *c, enum direction d) { return c->ngb[OppDir(d)]; }
  enum direction OppDir(enum direction d)
  {
    return static_cast<direction>((d % 2 == 0) ? d + 1 : d - 1);
  }
  double DX() const { return dx; }
  int idx() const { return 2; }
  size_t Ncell() const { return (size_t)ng[0] * ng[1] * ng[2]; }
  size_t Ncell_all() const { return (size_t)ncell; }
  double CellVolume(const cell *, const double) { return pow(dx, nd); }
  double CellInterface(const cell *, const direction, const double) { return pow(dx, nd - 1); }
  int boundary_depth(enum direction d) const { return nbc[d / 2]; }
  double DX(const cell *, const enum axes) const { return dx; }
  int Ndim() const { return nd; }
  int Nvar() const { return nv; }
  int NG(const enum axes a) const { return ng[a]; }
  int NG_All(const enum axes a) const { return nga[a]; }
  double Xmin(enum axes a) const { return xmin[a]; }
  double Xmax(enum axes a) const { return xmax[a]; }
  double Range(enum axes a) const { return xmax[a] - xmin[a]; }
  double Xmin_all(enum axes a) const { return xmin[a] - nbc[a] * dx; }
  double Xmax_all(enum axes a) const { return xmax[a] + nbc[a] * dx; }
  double Range_all(enum axes a) const { return Xmax_all(a) - Xmin_all(a); }
  double SIM_Xmin(enum axes a) const { return xmin[a]; }
  double SIM_Xmax(enum axes a) const { return xmax[a]; }
  double SIM_Range(enum axes a) const { return xmax[a] - xmin[a]; }
  double level_Xmin(enum axes a) const { return xmin[a]; }
  double level_Xmax(enum axes a) const { return xmax[a]; }
  double level_Range(enum axes a) const { return xmax[a] - xmin[a]; }
  int iXmin(enum axes) const { return 0; }
  int iXmax(enum axes a) const { return 2 * ng[a]; }
  int iRange(enum axes a) const { return 2 * ng[a]; }
  int iXmin_all(enum axes a) const { return -2 * nbc[a]; }
  int iXmax_all(enum axes a) const { return 2 * (ng[a] + nbc[a]); }
  int iRange_all(enum axes a) const { return 2 * nga[a]; }
  int SIM_iXmin(enum axes) const { return 0; }
  int SIM_iXmax(enum axes a) const { return 2 * ng[a]; }
  int SIM_iRange(enum axes a) const { return 2 * ng[a]; }
  int level_iXmin(enum axes) const { return 0; }
  int level_iXmax(enum axes a) const { return 2 * ng[a]; }
  int level_iRange(enum axes a) const { return 2 * ng[a]; }
  int SetupBCs(class SimParams &) { return 0; }
  int BC_printBCdata(boundary_data *) { return 0; }
  void BC_deleteBoundaryData() {}
  void BC_deleteBoundaryData(boundary_data *) {}
  double distance(const double *a, const double *b)
  {
    double t = 0;
    for (int i = 0; i < nd; i++) t += (a[i] - b[i]) * (a[i] - b[i]);
    return sqrt(t);
  }
  double distance_vertex2cell(const double *v, const cell *c)
  {
    double t = 0;
    for (int i = 0; i < nd; i++) {
      double d = v[i] - CI.get_dpos(c, i);
      t += d * d;
    }
    return sqrt(t);
  }
  double distance_cell2cell(const cell *a, const cell *b)
  {
    double t = 0;
    for (int i = 0; i < nd; i++) {
      double d = CI.get_dpos(a, i) - CI.get_dpos(b, i);
      t += d * d;
    }
    return sqrt(t);
  }
  double difference_vertex2cell(const double *v, const cell *c, const axes a)
  {
    return CI.get_dpos(c, a) - v[a];
  }
  double idistance(const int *a, const int *b)
  {
    double t = 0;
    for (int i = 0; i < nd; i++) t += double(a[i] - b[i]) * (a[i] - b[i]);
    return sqrt(t);
  }
  double idistance_cell2cell(const cell *a, const cell *b) { return idistance(a->pos, b->pos); }
  double idistance_vertex2cell(const int *v, const cell *c) { return idistance(v, c->pos); }
  double idifference_vertex2cell(const int *v, const cell *c, const axes a) { return c->pos[a] - v[a]; }
  double idifference_cell2cell(const cell *a, const cell *b, const axes x) { return b->pos[x] - a->pos[x]; }
  bool point_on_grid(const double *p)
  {
    for (int i = 0; i < nd; i++)
      if (p[i] < xmin[i] || p[i] > xmax[i]) return false;
    return true;
  }
};

// ---------------------------------------------------------------------------
// Microphysics object standing where mp_only_cooling stands: the temperature
// relations of mp_only_cooling.cpp:81-85,255-280 are two one-liners and are
// supplied here so that the REFERENCE's UtoP/sCMA/CellAdvanceTime MP branches
// can be exercised; sCMA itself is the reference's base-class code.
// (mp_only_cooling itself cannot be linked: its table sources need GSL.)
// ---------------------------------------------------------------------------
class HarnessMP : public microphysics_base {
 public:
  double Mu_tot_over_kB;
  HarnessMP(const int nv, const int ntr, const std::string *tr, struct which_physics *ep,
            struct rad_sources *rs)
      : microphysics_base(nv, ntr, tr, ep, rs)
  {
    Mu_tot_over_kB = 0.609 * pconst.m_p() / pconst.kB();
  }
  int TimeUpdateMP(const pion_flt *, pion_flt *, const double, const double, const int, double *)
  {
    return DONT_CALL_ME;
  }
  int TimeUpdateMP_RTnew(const pion_flt *, const int, const std::vector<struct rt_source_data> &,
                         const int, std::vector<struct rt_source_data> &, pion_flt *, const double,
                         const double, const int, double *)
  {
    return DONT_CALL_ME;
  }
  int TimeUpdate_RTsinglesrc(const pion_flt *, pion_flt *, const double, const double, const int,
                             const double, const double, const double, double *)
  {
    return DONT_CALL_ME;
  }
  int Init_ionfractions(pion_flt *, const double, const double) { return DONT_CALL_ME; }
  int Set_Temp(pion_flt *p, const double T, const double)
  {
    p[PG] = p[RO] * T / Mu_tot_over_kB;
    return 0;
  }
  double Temperature(const pion_flt *p, const double) { return p[PG] * Mu_tot_over_kB / p[RO]; }
  double timescales(const pion_flt *, const double, const bool, const bool, const bool) { return 1e99; }
  double timescales_RT(const pion_flt *, const int, const std::vector<struct rt_source_data> &,
                       const int, const std::vector<struct rt_source_data> &, const double)
  {
    return 1e99;
  }
  double get_recombination_rate(const int, const pion_flt *, const double) { return 0.0; }
  void get_dtau(const rad_source *, const pion_flt, const pion_flt *, pion_flt *) {}
  void get_ion_source_rate(const pion_flt *, const double, double *) {}
  double get_n_elec(const pion_flt *) { return 0.0; }
  double get_n_Hplus(const pion_flt *) { return 0.0; }
  double get_n_Hneutral(const pion_flt *) { return 0.0; }
  double get_n_ion(string, const pion_flt *) { return 0.0; }
  double get_X_H() { return 0.0; }
};

// Scalar ODE dE/dt = f(E) integrated by the REFERENCE's Cash-Karp integrator
// (microphysics/integrator.cpp); f is a piecewise-linear table handed in by the test.
class HarnessODE : public Integrator_Base {
 public:
  std::vector<double> xs, ys;
  HarnessODE() { Set_Nvar(1); }
  int dPdt(const int, const double *Pv, double *R)
  {
    const double x = Pv[0];
    size_t ihi = xs.size() - 1, ilo = 0, imid;
    do {
      imid = ilo + (ihi - ilo) / 2;
      if (xs[imid] < x) ilo = imid;
      else ihi = imid;
    } while (ihi - ilo > 1);
    R[0] = ys[ilo] + (x - xs[ilo]) * (ys[ilo + 1] - ys[ilo]) / (xs[ilo + 1] - xs[ilo]);
    return 0;
  }
  int C_rate(const int, const double *, double *) { return DONT_CALL_ME; }
  int D_rate(const int, const double *, double *) { return DONT_CALL_ME; }
};

// oracle/ref_cooling.cpp: the reference's own mp_only_cooling object (its three spline-backed rate
// curves are test doubles fed by the test); available once the test has supplied the curves
extern "C" int ref_cooling_have_curves();
extern "C" microphysics_base *ref_cooling_new_mp(int nv, int ntr, const std::string *tr, which_physics *ep,
                                                 rad_sources *rs);

// ---------------------------------------------------------------------------
struct RefSim : public periodic_bc,
                public oneway_out_bc,  // derives from outflow_bc
                public inflow_bc,
                public reflecting_bc,
                public fixed_bc,
                public double_Mach_ref_bc,
                public jet_bc,
                public axisymmetric_bc,
                public jetreflect_bc {
  pion_gpu_config cfg;
  SimParams par;
  HarnessGrid *grid;
  FV_solver_base *solver;
  microphysics_base *hmp;
  bool real_mp;  // MP is the reference's mp_only_cooling (else the HarnessMP stand-in)
  std::string trnames[PION_MAX_NVAR];
  cell *scratchL, *scratchR;

  static int ref_bc_type(int t)
  {
    switch (t) {
      case PION_BC_PERIODIC: return PERIODIC;
      case PION_BC_OUTFLOW: return OUTFLOW;
      case PION_BC_INFLOW: return INFLOW;
      case PION_BC_REFLECTING: return REFLECTING;
      case PION_BC_FIXED: return FIXED;
      case PION_BC_ONEWAY_OUT: return ONEWAY_OUT;
      case PION_BC_DMACH: return DMACH;
      case PION_BC_DMACH2: return DMACH2;
      case PION_BC_AXISYMMETRIC: return AXISYMMETRIC;
      case PION_BC_JETREFLECT: return JETREFLECT;
      default: return -1;
    }
  }

  RefSim(const pion_gpu_config &c) : cfg(c), grid(0), solver(0), hmp(0), real_mp(false)
  {
    par.gridType = 1;
    par.eqntype = c.eqntype;
    par.coord_sys = (c.coord_sys == 2) ? COORD_CYL : ((c.coord_sys == 3) ? COORD_SPH : COORD_CRT);
    par.solverType = c.solver;
    par.eqnNDim = 3;
    par.ndim = c.ndim;
    par.nvar = c.nvar;
    par.ntracer = c.ntracer;
    par.ftr = c.nvar - c.ntracer;
    par.simtime = 0.0;
    par.timestep = 0;
    par.dt = 0.0;
    par.dx = c.dx;
    par.grid_nlevels = 1;
    par.Nbc = c.nbc;
    par.spOOA = c.sp_ooa;
    par.tmOOA = c.tm_ooa;
    par.gamma = c.gamma;
    par.CFL = c.cfl;
    par.artviscosity = c.artvisc;
    par.etav = c.etav;
    par.EP.dynamics = 1;
    par.EP.raytracing = 0;
    par.EP.cooling = c.cooling;
    par.EP.chemistry = 0;
    par.EP.update_erg = 1;
    par.EP.MP_timestep_limit = c.mp_timestep_limit;
    par.EP.MinTemperature = c.min_temp;
    par.EP.MaxTemperature = c.max_temp;
    par.RS.Nsources = 0;
    for (int v = 0; v < c.nvar; v++) par.RefVec[v] = c.refvec[v];
    for (int a = 0; a < 3; a++) {
      par.NG[a] = (a < c.ndim) ? c.ng[a] : 1;
      par.Xmin[a] = c.xmin[a];
      par.Xmax[a] = c.xmin[a] + par.NG[a] * c.dx;
      par.Range[a] = par.Xmax[a] - par.Xmin[a];
    }
    CI.set_ndim(c.ndim);
    CI.set_nvar(c.nvar);
    CI.set_xmin(par.Xmin);
    CI.set_nlevels(c.dx, 1);
    CI.setup_extra_data(par.RS, c.ndim, 1, 1);
    grid = new HarnessGrid(c);
    scratchL = CI.new_cell();
    scratchR = CI.new_cell();

    // microphysics stand-in (see HarnessMP)
    MP = 0;
    if (c.cooling != 0) {
      for (int t = 0; t < c.ntracer; t++) trnames[t] = "colour";
      if (ref_cooling_have_curves()) {
        hmp = ref_cooling_new_mp(c.nvar, c.ntracer, trnames, &par.EP, &par.RS);
        real_mp = true;
      }
      else
        hmp = new HarnessMP(c.nvar, c.ntracer, trnames, &par.EP, &par.RS);
      MP = hmp;
    }
    // setup_fixed_grid::set_equations (grid/setup_fixed_grid.cpp:1067-1191), Cartesian
    pion_flt *rv = par.RefVec;
    if (c.coord_sys == 3) {
      // spherical symmetry, 1-D, hydro only (setup_fixed_grid.cpp:1161-1175)
      solver = new sph_FV_solver_Hydro_Euler(c.nvar, c.ndim, c.cfl, c.gamma, rv, c.etav, c.ntracer);
    }
    else if (c.coord_sys == 2) {
      // cylindrical (z,R) axisymmetry: setup_fixed_grid.cpp:1133-1160
      if (c.eqntype == EQEUL)
        solver = new cyl_FV_solver_Hydro_Euler(c.nvar, c.ndim, c.cfl, c.gamma, rv, c.etav, c.ntracer);
      else if (c.eqntype == EQMHD)
        solver = new cyl_FV_solver_mhd_ideal_adi(c.nvar, c.ndim, c.cfl, c.gamma, rv, c.etav, c.ntracer);
      else
        solver = new cyl_FV_solver_mhd_mixedGLM_adi(c.nvar, c.ndim, c.cfl, c.gamma, rv, c.etav, c.ntracer);
    }
    else if (c.eqntype == EQEUL)
      solver = new FV_solver_Hydro_Euler(c.nvar, c.ndim, c.cfl, c.gamma, rv, c.etav, c.ntracer);
    else if (c.eqntype == EQMHD)
      solver = new FV_solver_mhd_ideal_adi(c.nvar, c.ndim, c.cfl, c.gamma, rv, c.etav, c.ntracer);
    else
      solver = new FV_solver_mhd_mixedGLM_adi(c.nvar, c.ndim, c.cfl, c.gamma, rv, c.etav, c.ntracer);
    solver->SetEOS(c.gamma);
    setup_bcs();
  }
  ~RefSim()
  {
    delete solver;
    CI.delete_cell(scratchL);
    CI.delete_cell(scratchR);
    delete grid;
    if (hmp) delete hmp;
    MP = 0;
  }

  // boundary lists: UniformGrid::SetupBCs (grid/uniform_grid.cpp:1009-1216) +
  // setup of boundary_data (uniform_grid.cpp BC_setBCtypes)
  void setup_bcs()
  {
    const int nd = cfg.ndim;
    for (int d = 0; d < 2 * nd; d++) {
      boundary_data *b = new boundary_data;
      b->dir = static_cast<direction>(d);
      b->ondir = grid->OppDir(b->dir);
      b->baxis = static_cast<axes>(d / 2);
      b->bpos = (d % 2) == 1;
      b->itype = ref_bc_type(cfg.bc_type[d]);
      b->refval = 0;
      b->depth = cfg.nbc;
      grid->BC_bd.push_back(b);
    }
    const int *ng = grid->ng, *nbc = grid->nbc;
    for (int d = 0; d < 2 * nd; d++) {
      boundary_data *b = grid->BC_bd[d];
      const int ax = d / 2;
      const bool pos = d % 2;
      int lo[3], hi[3];
      for (int a = 0; a < 3; a++) {
        if (a < ax || a >= nd) {
          lo[a] = -nbc[a];
          hi[a] = ng[a] + nbc[a];
        }
        else {
          lo[a] = 0;
          hi[a] = ng[a];
        }
      }
      if (ax == 0) {
        for (int iz = lo[2]; iz < hi[2]; iz++)
          for (int iy = lo[1]; iy < hi[1]; iy++)
            for (int k = 0; k < nbc[0]; k++) {
              int ix = pos ? ng[0] + k : -nbc[0] + k;
              b->data.push_back(grid->cells[grid->id(ix, iy, iz)]);
            }
      }
      else if (ax == 1) {
        for (int iz = lo[2]; iz < hi[2]; iz++)
          for (int k = 0; k < nbc[1]; k++) {
            int iy = pos ? ng[1] + k : -nbc[1] + k;
            for (int ix = lo[0]; ix < hi[0]; ix++) b->data.push_back(grid->cells[grid->id(ix, iy, iz)]);
          }
      }
      else {
        for (int k = 0; k < nbc[2]; k++) {
          int iz = pos ? ng[2] + k : -nbc[2] + k;
          for (int iy = lo[1]; iy < hi[1]; iy++)
            for (int ix = lo[0]; ix < hi[0]; ix++) b->data.push_back(grid->cells[grid->id(ix, iy, iz)]);
        }
      }
    }
    if (cfg.bc_dmach2) {
      boundary_data *b = new boundary_data;
      b->dir = NO;
      b->ondir = NO;
      b->itype = DMACH2;
      b->refval = 0;
      grid->BC_bd.push_back(b);
    }
  }
  // jet simulation: JP (sim_params.h:331-341) + the internal JETBC boundary, which the reference puts
  // after the external ones in the list (uniform_grid.cpp BC_setBCtypes, "internal" boundaries)
  void set_jet(int radius, const double *state)
  {
    JP.jetic = 1;
    JP.jetradius = radius;
    for (int v = 0; v < MAX_NVAR; v++) JP.jetstate[v] = (v < cfg.nvar) ? state[v] : 0.0;
    boundary_data *b = new boundary_data;
    b->dir = NO;
    b->ondir = NO;
    b->itype = JETBC;
    b->refval = 0;
    grid->BC_bd.push_back(b);
  }
  // assign_update_bcs::assign_boundary_data (boundaries/assign_update_bcs.cpp:58-131)
  void assign_bcs()
  {
    for (size_t i = 0; i < grid->BC_bd.size(); i++) {
      boundary_data *b = grid->BC_bd[i];
      switch (b->itype) {
        case PERIODIC: BC_assign_PERIODIC(par, 0, grid, b); break;
        case OUTFLOW: BC_assign_OUTFLOW(par, grid, b); break;
        case ONEWAY_OUT: BC_assign_ONEWAY_OUT(par, grid, b); break;
        case INFLOW: BC_assign_INFLOW(par, grid, b); break;
        case REFLECTING: BC_assign_REFLECTING(par, grid, b); break;
        case AXISYMMETRIC: BC_assign_AXISYMMETRIC(par, grid, b); break;
        case JETREFLECT: BC_assign_JETREFLECT(par, grid, b); break;
        case FIXED: BC_assign_FIXED(par, grid, b); break;
        case DMACH: BC_assign_DMACH(par, grid, b); break;
        case DMACH2:
          if (b->refval) b->refval = mem.myfree(b->refval);
          b->data.clear();
          BC_assign_DMACH2(par, grid, b);
          break;
        case JETBC:
          if (b->refval) b->refval = mem.myfree(b->refval);
          b->data.clear();
          BC_assign_JETBC(par, grid, b);
          break;
        default: break;
      }
    }
  }
  // TimeUpdateExternalBCs (boundaries/assign_update_bcs.cpp:185-252)
  void update_bcs(double simtime, int cstep, int maxstep)
  {
    for (size_t i = 0; i < grid->BC_bd.size(); i++) {
      boundary_data *b = grid->BC_bd[i];
      switch (b->itype) {
        case PERIODIC: BC_update_PERIODIC(par, 0, grid, b, cstep, maxstep); break;
        case OUTFLOW: BC_update_OUTFLOW(par, grid, b, cstep, maxstep); break;
        case ONEWAY_OUT: BC_update_ONEWAY_OUT(par, grid, b, cstep, maxstep); break;
        case INFLOW: BC_update_INFLOW(par, grid, b, cstep, maxstep); break;
        case REFLECTING: BC_update_REFLECTING(par, grid, b, cstep, maxstep); break;
        case AXISYMMETRIC: BC_update_AXISYMMETRIC(par, grid, b, cstep, maxstep); break;
        case JETREFLECT: BC_update_JETREFLECT(par, grid, b, cstep, maxstep); break;
        case FIXED: BC_update_FIXED(par, grid, b, cstep, maxstep); break;
        case DMACH: BC_update_DMACH(par, grid, simtime, b, cstep, maxstep); break;
        case DMACH2: BC_update_DMACH2(par, grid, b, cstep, maxstep); break;
        case JETBC: BC_update_JETBC(par, grid, b, cstep, maxstep); break;
        default: break;
      }
    }
  }

  // time_integrator::dynamics_dU_column (sim_control/time_integrator.cpp:645-873),
  // every physics call goes to the reference's solver object.
  int dynamics_dU_column(cell *startingPt, const enum direction posdir, const enum direction negdir,
                         const double dt, const int csp)
  {
    int err = 0;
    enum axes axis = solver->GetDirection();
    double dx = grid->DX();
    const int nvar = par.nvar;
    std::vector<pion_flt> buf(6 * nvar, 0.0);
    pion_flt *Fr_prev = &buf[0], *Fr_this = &buf[nvar], *slope_cpt = &buf[2 * nvar],
             *slope_npt = &buf[3 * nvar], *edgeL = &buf[4 * nvar], *edgeR = &buf[5 * nvar], *temp = 0;
    cell *cpt = startingPt;
    cell *npt = grid->NextPt(cpt, posdir);
    cell *n2pt = grid->NextPt(npt, posdir);
    if (npt == 0 || n2pt == 0) rep.error("Couldn't find two real cells in column", 0);
    do {
      err += solver->SetEdgeState(cpt, posdir, nvar, slope_cpt, edgeL, csp, grid);
      err += solver->SetSlope(npt, axis, nvar, slope_npt, csp, grid);
      err += solver->SetEdgeState(npt, negdir, nvar, slope_npt, edgeR, csp, grid);
      err += solver->InterCellFlux(par, grid, cpt, npt, edgeL, edgeR, Fr_this, par.gamma, dx);
      err += solver->MHDsource(grid, cpt, npt, edgeL, edgeR, axis, posdir, negdir, dt);
      err += solver->dU_Cell(grid, cpt, axis, Fr_prev, Fr_this, slope_cpt, csp, dx, dt);
      temp = Fr_prev; Fr_prev = Fr_this; Fr_this = temp;
      temp = slope_cpt; slope_cpt = slope_npt; slope_npt = temp;
      cpt = npt;
      npt = n2pt;
    } while ((n2pt = grid->NextPt(n2pt, posdir)));
    err += solver->SetEdgeState(cpt, posdir, nvar, slope_cpt, edgeL, csp, grid);
    for (int v = 0; v < nvar; v++) slope_npt[v] = 0.;
    err += solver->SetEdgeState(npt, negdir, nvar, slope_npt, edgeR, csp, grid);
    err += solver->InterCellFlux(par, grid, cpt, npt, edgeL, edgeR, Fr_this, par.gamma, dx);
    err += solver->MHDsource(grid, cpt, npt, edgeL, edgeR, axis, posdir, negdir, dt);
    err += solver->dU_Cell(grid, cpt, axis, Fr_prev, Fr_this, slope_cpt, csp, dx, dt);
    return err;
  }
  // time_integrator::set_dynamics_dU (:553-636)
  int set_dynamics_dU(const double dt, const int step)
  {
    enum direction posdirs[3] = {XP, YP, ZP}, negdirs[3] = {XN, YN, ZN};
    enum axes axis[3] = {XX, YY, ZZ};
    int space_ooa = (step == OA1) ? OA1 : OA2;
    for (int i = 0; i < par.ndim; i++) {
      solver->SetDirection(axis[i]);
      class cell *cpt = grid->FirstPt_All();
      class cell *marker = cpt;
      enum direction d1 = posdirs[(i + 1) % 3];
      enum direction d2 = posdirs[(i + 2) % 3];
      enum axes x1 = axis[(i + 1) % 3];
      enum axes x2 = axis[(i + 2) % 3];
      for (int ax2 = 0; ax2 < grid->NG_All(x2); ax2++) {
        for (int ax1 = 0; ax1 < grid->NG_All(x1); ax1++) {
          dynamics_dU_column(cpt, posdirs[i], negdirs[i], dt, space_ooa);
          cpt = grid->NextPt(cpt, d1);
        }
        marker = grid->NextPt(marker, d2);
        cpt = marker;
      }
    }
    solver->SetDirection(axis[0]);
    return 0;
  }
  // time_integrator::grid_update_state_vector (:881-958)
  void grid_update_state_vector(const double dt, const int step, const int ooa)
  {
    pion_flt temperg = 0.0;
    class cell *c = grid->FirstPt_All();
    do {
      if (!c->isdomain || !c->isleaf) {
        for (int v = 0; v < par.nvar; v++) c->dU[v] = 0.0;
      }
      else {
        solver->CellAdvanceTime(c, c->P, c->dU, c->Ph, &temperg, par.gamma, par.EP.MinTemperature, dt);
      }
      if (MP) {
        double T = MP->Temperature(c->Ph, par.gamma);
        if (T > par.EP.MaxTemperature) MP->Set_Temp(c->Ph, par.EP.MaxTemperature, par.gamma);
      }
      if (step == ooa)
        for (int v = 0; v < par.nvar; v++) c->P[v] = c->Ph[v];
    } while ((c = grid->NextPt_All(c)) != 0);
  }
  // time_integrator::calc_noRT_microphysics_dU (sim_control/time_integrator.cpp:438-489): every call
  // goes to the reference's MP and solver objects
  int calc_noRT_microphysics_dU(const double delt)
  {
    cell *c = grid->FirstPt_All();
    std::vector<pion_flt> buf(3 * par.nvar);
    pion_flt *p = &buf[0], *ui = &buf[par.nvar], *uf = &buf[2 * par.nvar];
    double tt = 0.;
    int err = 0;
    do {
      if (c->isdomain) {
        err += MP->TimeUpdateMP(c->P, p, delt, par.gamma, 0, &tt);
        solver->PtoU(c->P, ui, par.gamma);
        solver->PtoU(p, uf, par.gamma);
        for (int v = 0; v < par.nvar; v++) c->dU[v] += uf[v] - ui[v];
      }
    } while ((c = grid->NextPt_All(c)) != 0);
    return err;
  }
  // calc_timestep::get_mp_timescales_no_radiation (sim_control/calc_timestep.cpp:405-507), limits 1-3
  // (all of them ask mp_only_cooling for the cooling time only, :445-455)
  double get_mp_timescales_no_radiation()
  {
    double tempdt = 0.0, dt = 1.0e99;
    class cell *c = grid->FirstPt();
    do {
      if (!(c->isbd || !c->isleaf)) {
        tempdt = MP->timescales(c->Ph, par.gamma, true, false, false);
        dt = min(dt, tempdt);
      }
    } while ((c = grid->NextPt(c)) != 0);
    return dt;
  }
  // first_order_update / second_order_update (:151-250); the microphysics dU only with the real MP object
  void stage(double dt, int space_ooa, int is_full)
  {
    solver->Setdt(dt);
    par.dt = dt;
    if (real_mp && par.EP.cooling) calc_noRT_microphysics_dU(dt);
    solver->preprocess_data(space_ooa, par, grid);
    set_dynamics_dU(dt, space_ooa);
    solver->PostProcess_dU(dt, space_ooa, par, grid);
    grid_update_state_vector(dt, is_full ? par.tmOOA : OA1, is_full ? par.tmOOA : -1);
  }
  void advance_time(double dt, double simtime)
  {
    if (par.tmOOA == OA1 && par.spOOA == OA1) {
      stage(dt, OA1, 1);
      update_bcs(simtime, OA1, OA1);
    }
    else {
      stage(0.5 * dt, OA1, 0);
      update_bcs(simtime, OA1, OA2);
      stage(dt, OA2, 1);
      update_bcs(simtime, OA2, OA2);
    }
  }
  // calc_timestep::calc_dynamics_dt (sim_control/calc_timestep.cpp:271-333)
  double calc_dynamics_dt()
  {
    double tempdt = 0.0, dt = 1.e100, dx = grid->DX();
    class cell *c = grid->FirstPt();
    do {
      if (c->timestep && !c->isbd) {
        tempdt = solver->CellTimeStep(c, par.gamma, dx);
        dt = min(dt, tempdt);
      }
      c = grid->NextPt(c);
    } while (c != 0);
    return dt;
  }
};

// ===================================================================== C API
extern "C" {

// for pion_amd/host/reference_bridge (the reference-side adapter drives this grid): SimParams / GridBaseClass
void *ref_simparams(void *h) { return &((RefSim *)h)->par; }
void *ref_grid(void *h) { return static_cast<GridBaseClass *>(((RefSim *)h)->grid); }

int ref_create(const pion_gpu_config *cfg, void **h)
{
  *h = new RefSim(*cfg);
  return 0;
}
void ref_destroy(void *h) { delete (RefSim *)h; }
long ref_ncell_all(void *h) { return ((RefSim *)h)->grid->ncell; }

int ref_upload(void *h, const double *Psoa)
{
  RefSim *s = (RefSim *)h;
  const long n = s->grid->ncell;
  for (long c = 0; c < n; c++)
    for (int v = 0; v < s->par.nvar; v++) {
      s->grid->cells[c]->P[v] = Psoa[v * n + c];
      s->grid->cells[c]->Ph[v] = Psoa[v * n + c];
      s->grid->cells[c]->dU[v] = 0.0;
    }
  return 0;
}
int ref_download(void *h, int which, double *Psoa)
{
  RefSim *s = (RefSim *)h;
  const long n = s->grid->ncell;
  for (long c = 0; c < n; c++)
    for (int v = 0; v < s->par.nvar; v++) {
      const cell *cc = s->grid->cells[c];
      Psoa[v * n + c] = (which == 0) ? cc->P[v] : (which == 1 ? cc->Ph[v] : cc->dU[v]);
    }
  return 0;
}
int ref_get_flags(void *h, unsigned char *out)
{
  RefSim *s = (RefSim *)h;
  for (long c = 0; c < s->grid->ncell; c++) {
    const cell *cc = s->grid->cells[c];
    out[c] = (cc->isgd ? PION_CELL_ISGD : 0) | (cc->isbd ? PION_CELL_ISBD : 0) |
             (cc->isdomain ? PION_CELL_ISDOMAIN : 0) | (cc->timestep ? PION_CELL_TIMESTEP : 0) |
             (cc->isleaf ? PION_CELL_ISLEAF : 0);
  }
  return 0;
}
int ref_get_aux(void *h, int which, double *out)
{
  RefSim *s = (RefSim *)h;
  for (long c = 0; c < s->grid->ncell; c++) {
    const cell *cc = s->grid->cells[c];
    if (which < 3) out[c] = (which < s->par.ndim) ? CI.get_Hcorr(cc, static_cast<axes>(which)) : 0.0;
    else if (which == 3) out[c] = CI.get_DivV(cc);
    else out[c] = CI.get_MagGradP(cc);
  }
  return 0;
}
int ref_update_bcs(void *h, double simtime, int cstep, int maxstep, int assign)
{
  RefSim *s = (RefSim *)h;
  s->par.simtime = simtime;
  if (assign) s->assign_bcs();
  s->update_bcs(simtime, cstep, maxstep);
  return 0;
}
int ref_set_jet(void *h, int radius, const double *state)
{
  ((RefSim *)h)->set_jet(radius, state);
  return 0;
}
int ref_calc_dt(void *h, double *t_dyn, double *t_mp)
{
  RefSim *s = (RefSim *)h;
  *t_dyn = s->calc_dynamics_dt();
  *t_mp = (s->real_mp && s->par.EP.MP_timestep_limit >= 1 && s->par.EP.MP_timestep_limit <= 3) ? s->get_mp_timescales_no_radiation() : 1.0e99;
  return 0;
}
int ref_set_glm_speeds(void *h, double dt, double dx, double cr)
{
  ((RefSim *)h)->solver->Set_GLM_Speeds(dt, dx, cr);
  return 0;
}
int ref_stage(void *h, double dt, int space_ooa, int is_full)
{
  ((RefSim *)h)->stage(dt, space_ooa, is_full);
  return 0;
}
int ref_setdt(void *h, double dt)
{
  ((RefSim *)h)->solver->Setdt(dt);
  return 0;
}
int ref_preprocess(void *h, int csp)
{
  RefSim *s = (RefSim *)h;
  return s->solver->preprocess_data(csp, s->par, s->grid);
}
int ref_set_dynamics_dU(void *h, double dt, int step) { return ((RefSim *)h)->set_dynamics_dU(dt, step); }
int ref_grid_update(void *h, double dt, int is_full)
{
  RefSim *s = (RefSim *)h;
  s->grid_update_state_vector(dt, is_full ? s->par.tmOOA : OA1, is_full ? s->par.tmOOA : -1);
  return 0;
}
int ref_advance_time(void *h, double dt, double simtime)
{
  ((RefSim *)h)->advance_time(dt, simtime);
  return 0;
}
// InterCellFlux on n independent interfaces (same seam as pion_gpu_interface_flux)
int ref_interface_flux(void *h, int n, int axis, double dt, const double *Pl, const double *Pr,
                       const double *aux, double *F, double *Pstar)
{
  RefSim *s = (RefSim *)h;
  const int nv = s->par.nvar;
  s->solver->SetDirection(static_cast<axes>(axis));
  s->solver->Setdt(dt);
  std::vector<pion_flt> l(nv), r(nv);
  for (int i = 0; i < n; i++) {
    for (int v = 0; v < nv; v++) {
      l[v] = Pl[(size_t)i * nv + v];
      r[v] = Pr[(size_t)i * nv + v];
    }
    for (int a = 0; a < s->par.ndim; a++) {
      CI.set_Hcorr(s->scratchL, static_cast<axes>(a), aux[4 * i]);
      CI.set_Hcorr(s->scratchR, static_cast<axes>(a), aux[4 * i]);
    }
    const bool hll = aux[4 * i + 1] != 0.0;
    CI.set_DivV(s->scratchL, hll ? -1.0 : 0.0);
    CI.set_MagGradP(s->scratchL, hll ? 10.0 : 0.0);
    CI.set_DivV(s->scratchR, 0.0);
    CI.set_MagGradP(s->scratchR, 0.0);
    s->solver->InterCellFlux(s->par, s->grid, s->scratchL, s->scratchR, &l[0], &r[0],
                             F + (size_t)i * nv, s->par.gamma, s->par.dx);
    (void)Pstar;
  }
  s->solver->SetDirection(XX);
  return 0;
}
int ref_cell_advance(void *h, int n, double fv_dt, const double *Pin, const double *dU, double *Pf)
{
  RefSim *s = (RefSim *)h;
  const int nv = s->par.nvar;
  s->solver->Setdt(fv_dt);
  std::vector<pion_flt> p(nv), d(nv), f(nv);
  pion_flt dE = 0.0;
  for (int i = 0; i < n; i++) {
    for (int v = 0; v < nv; v++) {
      p[v] = Pin[(size_t)i * nv + v];
      d[v] = dU[(size_t)i * nv + v];
    }
    s->solver->CellAdvanceTime(s->scratchL, &p[0], &d[0], &f[0], &dE, s->par.gamma,
                               s->par.EP.MinTemperature, fv_dt);
    for (int v = 0; v < nv; v++) Pf[(size_t)i * nv + v] = f[v];
  }
  return 0;
}
int ref_cell_timestep(void *h, int n, const double *Pin, double *dt)
{
  RefSim *s = (RefSim *)h;
  const int nv = s->par.nvar;
  for (int i = 0; i < n; i++) {
    for (int v = 0; v < nv; v++) s->scratchL->P[v] = Pin[(size_t)i * nv + v];
    dt[i] = s->solver->CellTimeStep(s->scratchL, s->par.gamma, s->par.dx);
  }
  return 0;
}
// Integrator_Base::Int_Adaptive_RKCK on dE/dt = pwlin(E)
int ref_integrate(int ntab, const double *xs, const double *ys, int n, const double *E0,
                  const double *dt, double errtol, double *Eout, double *tout, int *errs)
{
  HarnessODE ode;
  ode.xs.assign(xs, xs + ntab);
  ode.ys.assign(ys, ys + ntab);
  for (int i = 0; i < n; i++) {
    double e = E0[i], t = 0.0;
    errs[i] = ode.Int_Adaptive_RKCK(1, &e, 0.0, dt[i], errtol, &e, &t);
    Eout[i] = e;
    tout[i] = t;
  }
  return 0;
}
}  // extern "C"
