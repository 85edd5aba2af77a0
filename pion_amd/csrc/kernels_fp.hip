// kernels_fp.hip -- the floating-point kernels of the flux-update path for gfx950.
//
// Compiled twice (Makefile): -DPION_FPNS=fp_strict -ffp-contract=off and
// -DPION_FPNS=fp_fast -ffp-contract=fast, and per equation set (-DPION_EQSEL).
//
//   k_stage    one fused stage of the time integrator for every on-grid cell:
//              cooling source (time_integrator.cpp:438-489), the directionally-unsplit sweeps
//              (set_dynamics_dU/dynamics_dU_column, time_integrator.cpp:553-873) and the
//              state update (grid_update_state_vector + CellAdvanceTime, :881-958).
//              The reference accumulates cell->dU in memory sweep by sweep; here one
//              thread owns one cell, rebuilds the two interface fluxes of each axis from
//              a 5-point stencil per axis, and applies the contributions to its dU in
//              registers in exactly the reference's order (x,y,z; per axis: Powell/GLM term
//              of the lower interface, of the upper interface, then the flux difference).
//              HBM sees: read the stencil state (cached re-reads), read the start-of-step
//              state, write the new state.
//   k_prepass  div v / |grad p|/p HLL switch (solver_eqn_base.cpp:398-412) and H-correction
//              eta (calc_Hcorrection :423-570).
//   k_dt       CellTimeStep + cooling-time reduction (calc_timestep.cpp:271-507).
#include "dev_cooling.h"
#include "kernels.h"

#ifndef PION_FPNS
#error "PION_FPNS must be defined (fp_strict or fp_fast)"
#endif
#ifndef PION_EQSEL
#define PION_EQSEL 0  // 0 = shared kernels (prepass, dt, cooling tests); 1,2,3 = stage kernels per eqn
#endif

namespace pion {
namespace PION_FPNS {

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------
// SoA variable of sweep-frame slot s for a sweep along ax (run-time version of gvar<>)
template <bool MHD>
PDEV int rotvar(const int ax, const int s)
{
  if (s >= 2 && s <= 4) {
    int k = (s - 2) + ax;
    return 2 + (k >= 3 ? k - 3 : k);
  }
  if (MHD && s >= 5 && s <= 7) {
    int k = (s - 5) + ax;
    return 5 + (k >= 3 ? k - 3 : k);
  }
  return s;
}
template <int EQ, int NV>
PDEV void apply_axis(double *d, const double *q0, const double bnm, const double sim, const double bnp,
                     const double sip, const double *Fm, const double *Fp, const double dt, const double dx);
// BaseVectorOps::AvgFalle with AVG_MINMOD (coord_sys/VectorOps.cpp:37-59)
PDEV double avg_falle(const double a, const double b)
{
  if (a * b <= PION_VERY_TINY_VALUE) return 0.0;
#ifdef PION_FAST_MATH
  // a and b have the same sign here, so r = a/b > 0 and min(r,1)*b is a (if |a|<|b|) or b:
  // the fast build picks it without the division (differs from the reference form by <= 1 ulp)
  return (fabs(a) < fabs(b)) ? a : b;
#else
  double r = a / b;
  return (r > 0.0) ? dmin(r, 1.0) * b : 0.0;
#endif
}
// Cylindrical (z,R) grids: centre of a cell along R from its all-cell y index (cell_interface.cpp:506-512)
// and VectorOps_Cyl::R_com (coord_sys/VectorOps.h:414-418)
PDEV double cyl_R(const GridDesc &g, const int jy_all)
{
  return g.xmin[1] + (2 * (jy_all - g.nbc[1]) + 1) * (0.5 * g.dx);
}
PDEV double cyl_Rcom(const double R, const double dR) { return (R + dR * dR / 12. / R); }
// spherical symmetry (1-D): cell centre from the all-cell x index; R3 and R_com of VectorOps_Sph
// (coord_sys/VectorOps_spherical.h:172-196)
PDEV double sph_R(const GridDesc &g, const int ix_all)
{
  return g.xmin[0] + (2 * (ix_all - g.nbc[0]) + 1) * (0.5 * g.dx);
}
PDEV double sph_R3(const double R, const double dR) { return (R + dR * dR / 12.0 / R); }
PDEV double sph_Rcom(const double R, const double dR)
{
  double delta2 = dR / R;
  delta2 *= delta2;
  return R * (1.0 + 0.25 * delta2) / (1.0 + delta2 / 12.0);
}

// XCD-aware tile decode: workgroups are dealt round-robin to the 8 XCDs (b % 8 share an XCD);
// give each XCD a contiguous range of tiles so that the halo re-reads of neighbouring tiles hit
// the same L2.  Placement only changes speed, never results.
PDEV long xcd_tile(const long b, const long ntiles)
{
  const long chunk = (ntiles + 7) / 8;
  return (b % 8) * chunk + (b / 8);
}

// rotate the three vector components between the lab frame and the sweep frame
template <int NV, bool MHD>
PDEV void to_sweep(const int ax, const double *lab, double *sw)
{
#pragma unroll
  for (int v = 0; v < NV; v++) sw[v] = lab[v];
  if (ax == 1) {
    sw[2] = lab[3]; sw[3] = lab[4]; sw[4] = lab[2];
    if constexpr (MHD) { sw[5] = lab[6]; sw[6] = lab[7]; sw[7] = lab[5]; }
  }
  else if (ax == 2) {
    sw[2] = lab[4]; sw[3] = lab[2]; sw[4] = lab[3];
    if constexpr (MHD) { sw[5] = lab[7]; sw[6] = lab[5]; sw[7] = lab[6]; }
  }
}
template <int NV, bool MHD>
PDEV void from_sweep(const int ax, const double *sw, double *lab)
{
#pragma unroll
  for (int v = 0; v < NV; v++) lab[v] = sw[v];
  if (ax == 1) {
    lab[3] = sw[2]; lab[4] = sw[3]; lab[2] = sw[4];
    if constexpr (MHD) { lab[6] = sw[5]; lab[7] = sw[6]; lab[5] = sw[7]; }
  }
  else if (ax == 2) {
    lab[4] = sw[2]; lab[2] = sw[3]; lab[3] = sw[4];
    if constexpr (MHD) { lab[7] = sw[5]; lab[5] = sw[6]; lab[6] = sw[7]; }
  }
}

#include "dev_addr.h"

// CellTimeStep of a lab-frame state (solver_eqn_hydro_adi.cpp:460-502 / solver_eqn_mhd_adi.cpp:516-582);
// the same operations as k_dt, used by the stage kernel to leave the next step's dt behind.
template <int EQ>
PDEV double cell_dt(const double *P, const int ndim, const double g, const double dx, const double cfl)
{
  double p[8];
#pragma unroll
  for (int v = 0; v < 8; v++) p[v] = (EQ != EQEUL || v < 5) ? P[v] : 0.0;
  double temp;
  if constexpr (EQ == EQEUL) {
#ifdef PION_FAST_MATH
    // fast build: seeded root (of |v|^2 + 1e-200: a cell at rest would otherwise ask for the root of zero),
    // one reciprocal for dx / (|v| + c); multiply-adds written out (see below)
    temp = PION_VERY_TINY_VALUE;
    for (int v = 0; v < ndim; v++) temp = __builtin_fma(p[2 + v], p[2 + v], temp);
    temp = sqrt_pos(temp) + sqrt_pos((g * p[1]) * frcp(p[0]));
    return (dx * cfl) * frcp(temp);
#else
    temp = 0.0;
    for (int v = 0; v < ndim; v++) temp += p[2 + v] * p[2 + v];
    temp = sqrt(temp);
    temp += Eqn<EQEUL, 0>::chydro(p, g);
#endif
  }
#ifdef PION_FAST_MATH
  else if (ndim > 1) {
    // fast build: the fast speed along the weakest-field axis needs rho, p, |B|^2 and the smallest of the three
    // B_i^2 only -- no rotation of the state into that axis (the reference's sum of squares runs in the rotated
    // order: last-bit differences), shared reciprocal, seeded roots
    // (multiply-adds written out: this function is compiled into k_dt and into the stage kernels, and a restart
    // continues bit for bit only if both contract the same way)
    temp = fmx(fabs(p[2]), fabs(p[3]));
    if (ndim > 2) temp = fmx(temp, fabs(p[4]));
    const double bx2 = p[5] * p[5], by2 = p[6] * p[6], bz2 = p[7] * p[7];
    const double bn2 = fmn(fmn(bx2, by2), bz2);
    const double ir = frcp(p[0]);
    const double a2 = (g * p[1]) * ir;
    const double t1 = __builtin_fma((bx2 + by2) + bz2, ir, a2);
    const double t2 = fmx(PION_MACHINEACCURACY, __builtin_fma(t1, t1, -(4. * a2) * (bn2 * ir)));
    temp += sqrt_pos((t1 + sqrt_pos(t2)) * 0.5);
    return (dx * cfl) * frcp(temp);
  }
#endif
  else {
    temp = fabs(p[2]);
    if (ndim > 1) temp = dmax(temp, fabs(p[3]));
    if (ndim > 2) temp = dmax(temp, fabs(p[4]));
    if (ndim == 1) temp += Eqn<EQMHD, 0>::cfast(p, g);
    else {
      int newdir = 0;
      if (fabs(p[6]) < fabs(p[5])) {
        newdir = 1;
        if (fabs(p[7]) < fabs(p[6])) newdir = 2;
      }
      else if (fabs(p[7]) < fabs(p[5])) newdir = 2;
      double u1[8];
      to_sweep<8, true>(newdir, p, u1);
      temp += Eqn<EQMHD, 0>::cfast(u1, g);
    }
  }
  double t = dx / temp;
  t *= cfl;
  return t;
}

#if PION_EQSEL != 0
// ---------------------------------------------------------------------------
// select_Hcorr_eta (solver_eqn_base.cpp:608-678) for the interface (cl | cl+st) along ax.
// The "negative direction" look-ups step back along the SWEEP axis (negdir = 2*axis, :661),
// as in the reference.
// ---------------------------------------------------------------------------
PDEV double select_hcorr_eta(const StageArgs &a, const int ax, const long cl, const long st)
{
  const long nc = a.g.ncell;
  const long cr = cl + st;
  const int nd = a.g.ndim;
  double eta = a.eta[ax * nc + cl];
  if (nd == 1) return eta;
  int perp = (ax + 1) % nd;
  eta = dmax(eta, a.eta[perp * nc + cl]);
  eta = dmax(eta, a.eta[perp * nc + cr]);
  if (nd > 2) {
    perp = (ax + 2) % nd;
    eta = dmax(eta, a.eta[perp * nc + cl]);
    eta = dmax(eta, a.eta[perp * nc + cr]);
  }
  for (int idim = 1; idim < nd; idim++) {
    perp = (ax + idim) % nd;
    // cl-st and cr-st exist for every interface an on-grid cell needs (nbc >= 2)
    eta = dmax(eta, a.eta[perp * nc + (cl - st)]);
    eta = dmax(eta, a.eta[perp * nc + (cr - st)]);
  }
  return eta;
}

// ---------------------------------------------------------------------------
// the fused stage kernel
// ---------------------------------------------------------------------------
template <int EQ, int NTR, int SOLVER>
__global__ __launch_bounds__(256) void k_stage(const StageArgs a)
{
  typedef Eqn<EQ, NTR> E;
  typedef Flux<EQ, NTR, SOLVER> FX;
  constexpr int NV = E::NV;
  constexpr int BASE = E::BASE;
  constexpr bool MHD = E::MHD;

  // ---- which cell -------------------------------------------------------
  const int nbx = (a.g.ng[0] + 63) / 64, nby = (a.g.ng[1] + 3) / 4;
  const long ntiles = (long)nbx * nby * a.g.ng[2];
  const long tile = xcd_tile(blockIdx.x, ntiles);
  if (tile >= ntiles) return;
  const int bx = (int)(tile % nbx), by = (int)((tile / nbx) % nby), bz = (int)(tile / ((long)nbx * nby));
  const int ix = bx * 64 + (threadIdx.x & 63), iy = by * 4 + (threadIdx.x >> 6), iz = bz;
  if (ix >= a.g.ng[0] || iy >= a.g.ng[1]) return;
  const long nc = a.g.ncell;
  const long c = (long)(ix + a.g.nbc[0]) + a.g.sy * (iy + a.g.nbc[1]) + a.g.sz * (iz + a.g.nbc[2]);
  const double g = a.fc.gamma, dx = a.g.dx, dt = a.dt;
  int err = 0;

  // start-of-step state of this cell (lab frame)
  double P0[NV];
#pragma unroll
  for (int v = 0; v < NV; v++) P0[v] = a.Pc[v * nc + c];

  const uint8_t fl = a.flags[c];
  if (!(fl & 4 /*ISDOMAIN*/) || !(fl & 16 /*ISLEAF*/)) {
    // grid_update_state_vector skips the cell (time_integrator.cpp:905-908): Ph keeps its value
#pragma unroll
    for (int v = 0; v < NV; v++) a.out[v * nc + c] = P0[v];
    return;
  }

  double dU[NV];
#pragma unroll
  for (int v = 0; v < NV; v++) dU[v] = 0.0;

  // ---- microphysics source (calc_noRT_microphysics_dU) --------------------
  if (a.cooling != 0) {
    const double pg_new = Cooling::time_update(a.cool, P0[qRO], P0[qPG], dt, g, err);
    double pn[NV], ui[NV], uf[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) pn[v] = P0[v];
    pn[qPG] = pg_new;
    E::PtoU(P0, ui, g);
    E::PtoU(pn, uf, g);
#pragma unroll
    for (int v = 0; v < NV; v++) dU[v] += uf[v] - ui[v];
  }

  // ---- sweeps -------------------------------------------------------------
  const bool oa2 = (a.space_ooa == 2);
#pragma unroll 1
  for (int ax = 0; ax < a.g.ndim; ax++) {
    const long st = (ax == 0) ? 1 : ((ax == 1) ? a.g.sy : a.g.sz);
    // R sweep of a cylindrical (z,R) grid: geometry enters slopes, edge states, the flux divergence and
    // the source terms (VectorOps_Cyl, cyl_FV_solver_*); jy = all-cell y index of this cell
    const bool cylR = (a.g.cyl == 1 && ax == 1);
    const bool sphR = (a.g.cyl == 2 && ax == 0);   // spherical symmetry: the only axis is R
    const int jy = sphR ? ix + a.g.nbc[0] : iy + a.g.nbc[1];
    double Rm1 = 0.0, R0 = 0.0, Rp1 = 0.0;
    if (cylR) {
      Rm1 = cyl_R(a.g, jy - 1);
      R0 = cyl_R(a.g, jy);
      Rp1 = cyl_R(a.g, jy + 1);
    }
    else if (sphR) {
      Rm1 = sph_R(a.g, jy - 1);
      R0 = sph_R(a.g, jy);
      Rp1 = sph_R(a.g, jy + 1);
    }
    double qm1[NV], q0[NV], qp1[NV], sm1[NV], s0[NV], sp1[NV];
    {
      double qm2[NV], qp2[NV];
#pragma unroll
      for (int s = 0; s < NV; s++) {
        const double *b = a.S + (long)rotvar<MHD>(ax, s) * nc + c;
        qm1[s] = b[-st];
        q0[s] = b[0];
        qp1[s] = b[st];
        if (oa2) {
          qm2[s] = b[-2 * st];
          qp2[s] = b[2 * st];
        }
        else {
          qm2[s] = qp2[s] = 0.0;
        }
      }
      // SetSlope (VectorOps.cpp:578-617; along R of a cylindrical grid VectorOps_Cyl::SetSlope
      // :1103-1204: differences of the cells' centres of mass)
      if (cylR || sphR) {
        // (VectorOps_Sph::SetSlope, VectorOps_spherical.cpp:358-440, has the same form with its own R_com)
        const double c_m2 = sphR ? sph_Rcom(sph_R(a.g, jy - 2), dx) : cyl_Rcom(cyl_R(a.g, jy - 2), dx),
                     c_m1 = sphR ? sph_Rcom(Rm1, dx) : cyl_Rcom(Rm1, dx),
                     c_0 = sphR ? sph_Rcom(R0, dx) : cyl_Rcom(R0, dx),
                     c_p1 = sphR ? sph_Rcom(Rp1, dx) : cyl_Rcom(Rp1, dx),
                     c_p2 = sphR ? sph_Rcom(sph_R(a.g, jy + 2), dx) : cyl_Rcom(cyl_R(a.g, jy + 2), dx);
#pragma unroll
        for (int s = 0; s < NV; s++) {
          if (oa2) {
            sm1[s] = avg_falle((qm1[s] - qm2[s]) / (c_m1 - c_m2), (q0[s] - qm1[s]) / (c_0 - c_m1));
            s0[s] = avg_falle((q0[s] - qm1[s]) / (c_0 - c_m1), (qp1[s] - q0[s]) / (c_p1 - c_0));
            sp1[s] = avg_falle((qp1[s] - q0[s]) / (c_p1 - c_0), (qp2[s] - qp1[s]) / (c_p2 - c_p1));
          }
          else {
            sm1[s] = s0[s] = sp1[s] = 0.0;
          }
        }
      }
      else {
#pragma unroll
        for (int s = 0; s < NV; s++) {
          if (oa2) {
            sm1[s] = avg_falle((qm1[s] - qm2[s]) / dx, (q0[s] - qm1[s]) / dx);
            s0[s] = avg_falle((q0[s] - qm1[s]) / dx, (qp1[s] - q0[s]) / dx);
            sp1[s] = avg_falle((qp1[s] - q0[s]) / dx, (qp2[s] - qp1[s]) / dx);
          }
          else {
            sm1[s] = s0[s] = sp1[s] = 0.0;
          }
        }
      }
    }
    // two interfaces: face 0 = (c-st | c), face 1 = (c | c+st)
    double Fm[NV], Fp[NV];
#pragma unroll 1
    for (int face = 0; face < 2; face++) {
      double eL[NV], eR[NV], f[NV], pstar[NV];
      // SetEdgeState (VectorOps.cpp:535-571)
#pragma unroll
      for (int s = 0; s < NV; s++) {
        const double ql = face ? q0[s] : qm1[s], sl = face ? s0[s] : sm1[s];
        const double qr = face ? qp1[s] : q0[s], sr = face ? sp1[s] : s0[s];
        if (oa2 && cylR) {
          // VectorOps_Cyl::SetEdgeState (VectorOps.cpp:1052-1092): distance of the face from the centre of mass
          const double Rl = face ? R0 : Rm1, Rr = face ? Rp1 : R0;
          eL[s] = ql + sl * (Rl + dx * 0.5 - cyl_Rcom(Rl, dx));
          eR[s] = qr + sr * (Rr - dx * 0.5 - cyl_Rcom(Rr, dx));
        }
        else if (oa2 && sphR) {
          // VectorOps_Sph::SetEdgeState (VectorOps_spherical.cpp:294-350)
          const double Rl = face ? R0 : Rm1, Rr = face ? Rp1 : R0;
          eL[s] = ql + sl * (Rl + 0.5 * dx - sph_Rcom(Rl, dx));
          eR[s] = qr + sr * (Rr - 0.5 * dx - sph_Rcom(Rr, dx));
        }
        else if (oa2) {
          eL[s] = ql + sl * dx * 0.5;
          eR[s] = qr - sr * dx * 0.5;
        }
        else {
          eL[s] = ql;
          eR[s] = qr;
        }
      }
      const long cl = face ? c : c - st;
      double hc_eta = 0.0;
      if (a.fc.artvisc == AV_HCORRECTION || a.fc.artvisc == AV_HCORR_FKJ98) hc_eta = select_hcorr_eta(a, ax, cl, st);
      bool use_hll = false;
      if constexpr (MHD && SOLVER == FLUX_RS_HLLD) use_hll = (a.hllflag[cl] | a.hllflag[cl + st]) != 0;
      FX::intercell_flux(eL, eR, f, pstar, a.fc, hc_eta, use_hll, err);
#pragma unroll
      for (int s = 0; s < NV; s++) {
        if (face == 0) Fm[s] = f[s];
        else Fp[s] = f[s];
      }
    }
    // accumulate in the sweep frame, in the reference's order
    double d[NV];
    to_sweep<NV, MHD>(ax, dU, d);
#ifdef PION_FAST_MATH
    // fast build, Cartesian axis: the same regrouped source + flux-difference update as k_stage_rows2
    // (apply_axis), so that the two kernels agree to rounding
    const bool fast_cart = !cylR && !sphR;
#else
    const bool fast_cart = false;
#endif
    if (fast_cart) {
      double bnm_ = 0.0, bnp_ = 0.0, sim_ = 0.0, sip_ = 0.0;
      if constexpr (MHD) {
        bnm_ = qm1[qBN];
        bnp_ = qp1[qBN];
      }
      if constexpr (EQ == EQGLM) {
        sim_ = qm1[qSI];
        sip_ = qp1[qSI];
      }
      apply_axis<EQ, NV>(d, q0, bnm_, sim_, bnp_, sip_, Fm, Fp, dt, dx);
    }
    else {
    if constexpr (MHD) {
      // MHDsource (solver_eqn_mhd_adi.cpp:396-443, GLM :782-813): this cell is the right cell
      // of face 0 and the left cell of face 1
      const double uB = q0[qBN] * q0[qVN] + q0[qBT1] * q0[qVT1] + q0[qBT2] * q0[qVT2];
      const double bm0 = 0.5 * (qm1[qBN] + q0[qBN]);
      if (cylR) {
        // cyl_FV_solver_mhd_ideal_adi::MHDsource, Rcyl (solver_eqn_mhd_adi.cpp:1081-1092): this cell as
        // the right cell of its lower face ...
        double rp = Rm1 + dx * 0.5;
        const double rn = rp;
        rp += dx;
        d[uMN] += dt * bm0 * (q0[qBN]) * 2.0 * rn / (rp * rp - rn * rn);
        d[uMT1] += dt * bm0 * (q0[qBT1]) * 2.0 * rn / (rp * rp - rn * rn);
        d[uMT2] += dt * bm0 * (q0[qBT2]) * 2.0 * rn / (rp * rp - rn * rn);
        d[uERG] += dt * bm0 * (uB) * 2.0 * rn / (rp * rp - rn * rn);
        d[uBN] += dt * bm0 * (q0[qVN]) * 2.0 * rn / (rp * rp - rn * rn);
        d[uBT1] += dt * bm0 * (q0[qVT1]) * 2.0 * rn / (rp * rp - rn * rn);
        d[uBT2] += dt * bm0 * (q0[qVT2]) * 2.0 * rn / (rp * rp - rn * rn);
      }
      else {
        d[uMN] += dt * bm0 * (q0[qBN]) / dx;
        d[uMT1] += dt * bm0 * (q0[qBT1]) / dx;
        d[uMT2] += dt * bm0 * (q0[qBT2]) / dx;
        d[uERG] += dt * bm0 * (uB) / dx;
        d[uBN] += dt * bm0 * (q0[qVN]) / dx;
        d[uBT1] += dt * bm0 * (q0[qVT1]) / dx;
        d[uBT2] += dt * bm0 * (q0[qVT2]) / dx;
      }
      if constexpr (EQ == EQGLM) {
        const double sm0 = 0.5 * (qm1[qSI] + q0[qSI]);
        d[uERG] += dt * sm0 * (q0[qVN] * q0[qSI]) / dx;
        d[uPSI] += dt * sm0 * q0[qVN] / dx;
      }
      const double bm1 = 0.5 * (q0[qBN] + qp1[qBN]);
      if (cylR) {
        // ... and as the left cell of its upper face
        const double rp = R0 + dx * 0.5;
        const double rn = rp - dx;
        d[uMN] -= dt * bm1 * (q0[qBN]) * 2.0 * rp / (rp * rp - rn * rn);
        d[uMT1] -= dt * bm1 * (q0[qBT1]) * 2.0 * rp / (rp * rp - rn * rn);
        d[uMT2] -= dt * bm1 * (q0[qBT2]) * 2.0 * rp / (rp * rp - rn * rn);
        d[uERG] -= dt * bm1 * (uB) * 2.0 * rp / (rp * rp - rn * rn);
        d[uBN] -= dt * bm1 * (q0[qVN]) * 2.0 * rp / (rp * rp - rn * rn);
        d[uBT1] -= dt * bm1 * (q0[qVT1]) * 2.0 * rp / (rp * rp - rn * rn);
        d[uBT2] -= dt * bm1 * (q0[qVT2]) * 2.0 * rp / (rp * rp - rn * rn);
      }
      else {
        d[uMN] -= dt * bm1 * (q0[qBN]) / dx;
        d[uMT1] -= dt * bm1 * (q0[qBT1]) / dx;
        d[uMT2] -= dt * bm1 * (q0[qBT2]) / dx;
        d[uERG] -= dt * bm1 * (uB) / dx;
        d[uBN] -= dt * bm1 * (q0[qVN]) / dx;
        d[uBT1] -= dt * bm1 * (q0[qVT1]) / dx;
        d[uBT2] -= dt * bm1 * (q0[qVT2]) / dx;
      }
      if constexpr (EQ == EQGLM) {
        const double sm1g = 0.5 * (q0[qSI] + qp1[qSI]);
        d[uERG] -= dt * sm1g * (q0[qVN] * q0[qSI]) / dx;
        d[uPSI] -= dt * sm1g * q0[qVN] / dx;
      }
    }
    // dU_Cell + DivStateVectorComponent (VectorOps.cpp:624-644)
    if (sphR) {
      // VectorOps_Sph::DivStateVectorComponent (VectorOps_spherical.cpp:449-475; the cell's shell volume
      // (rp^3-rn^3)/3 comes from the host, where pow() is the reference's) and
      // sph_FV_solver_Hydro_Euler::geometric_source (solver_eqn_hydro_adi.cpp:648-675)
      double u1[NV];
      const double rp = R0 + 0.5 * dx;
      const double rn = rp - dx;
      const double rc = a.g.sph_vol[jy];
#pragma unroll
      for (int s = 0; s < NV; s++) u1[s] = (rn * rn * Fm[s] - rp * rp * Fp[s]) / rc;
      if (oa2) u1[uMN] += 2.0 * ((q0[qPG] - s0[qPG] * sph_Rcom(R0, dx)) / sph_R3(R0, dx) + s0[qPG]);
      else u1[uMN] += 2.0 * q0[qPG] / sph_R3(R0, dx);
#pragma unroll
      for (int s = 0; s < NV; s++) d[s] += dt * u1[s];
    }
    else if (cylR) {
      // VectorOps_Cyl::DivStateVectorComponent, Rcyl (VectorOps.cpp:1211-1245) + geometric_source of the
      // cyl_FV_solver_* classes (solver_eqn_hydro_adi.cpp:560-590, solver_eqn_mhd_adi.cpp:1001-1030, 1175-1210)
      double u1[NV];
      const double rp = R0 + dx * 0.5;
      const double rn = rp - dx;
#pragma unroll
      for (int s = 0; s < NV; s++) u1[s] = 2.0 * (rn * Fm[s] - rp * Fp[s]) / (rp * rp - rn * rn);
      const double Rc = cyl_Rcom(R0, dx);
      if constexpr (!MHD) {
        if (oa2) u1[uMN] += (q0[qPG] + (R0 - Rc) * s0[qPG]) / R0;
        else u1[uMN] += q0[qPG] / R0;
      }
      else {
        const double pm = (q0[qBN] * q0[qBN] + q0[qBT1] * q0[qBT1] + q0[qBT2] * q0[qBT2]) / 2.;
        if (oa2) {
          u1[uMN] += (q0[qPG] + pm +
                      (R0 - Rc) * (s0[qPG] + q0[qBN] * s0[qBN] + q0[qBT1] * s0[qBT1] + q0[qBT2] * s0[qBT2])) /
                     R0;
          if constexpr (EQ == EQGLM) u1[uBN] += a.fc.chyp * (q0[qSI] + (R0 - Rc) * s0[qSI]) / R0;
        }
        else {
          u1[uMN] += (q0[qPG] + pm) / R0;
          if constexpr (EQ == EQGLM) u1[uBN] += a.fc.chyp * q0[qSI] / R0;
        }
      }
#pragma unroll
      for (int s = 0; s < NV; s++) d[s] += dt * u1[s];
    }
    else {
#pragma unroll
      for (int s = 0; s < NV; s++) {
        const double u1 = (Fm[s] - Fp[s]) / dx;
        d[s] += dt * u1;
      }
    }
    }
    from_sweep<NV, MHD>(ax, d, dU);
  }

  // ---- CellAdvanceTime (solver_eqn_hydro_adi.cpp:372-448, solver_eqn_mhd_adi.cpp:452-504, :822-844)
  double u1[NV], Pf[NV];
  if (a.fc.mp.present) {
    double Pi[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) Pi[v] = P0[v];
    E::apply_sCMA(Pi);
    E::PtoU(Pi, u1, g);
  }
  else E::PtoU(P0, u1, g);
#pragma unroll
  for (int v = 0; v < NV; v++) u1[v] += dU[v];
  E::UtoP(u1, Pf, a.fc.min_temp, g, a.fc.mp, err);
  if (a.fc.mp.present) E::apply_sCMA(Pf);
  if constexpr (EQ == EQGLM) Pf[qSI] *= a.glm_damp;  // GLMsource (eqns_mhd_adiabatic.cpp:651-660)
  if (a.fc.mp.present) {
    // grid_update_state_vector: T > MaxTemperature clamp (time_integrator.cpp:926-932)
    const double T = Pf[qPG] * a.fc.mp.Mu_tot_over_kB / Pf[qRO];
    if (T > a.max_temp) Pf[qPG] = Pf[qRO] * a.max_temp / a.fc.mp.Mu_tot_over_kB;
  }
#pragma unroll
  for (int v = 0; v < NV; v++) a.out[v * nc + c] = Pf[v];
  if (err) atomicOr(a.errword, err);
}

// ---------------------------------------------------------------------------
// interface-flux test seam: FV_solver_base::InterCellFlux on n independent interfaces
// ---------------------------------------------------------------------------
template <int EQ, int NTR, int SOLVER>
__global__ __launch_bounds__(256) void k_flux_test(const FluxTestArgs a)
{
  typedef Eqn<EQ, NTR> E;
  typedef Flux<EQ, NTR, SOLVER> FX;
  constexpr int NV = E::NV;
  constexpr bool MHD = E::MHD;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  double l[NV], r[NV], ls[NV], rs[NV], f[NV], ps[NV], fl[NV], pl[NV];
#pragma unroll
  for (int v = 0; v < NV; v++) {
    l[v] = a.Pl[(long)i * NV + v];
    r[v] = a.Pr[(long)i * NV + v];
  }
  to_sweep<NV, MHD>(a.axis, l, ls);
  to_sweep<NV, MHD>(a.axis, r, rs);
  int err = 0;
  FX::intercell_flux(ls, rs, f, ps, a.fc, a.aux[4 * i + 0], a.aux[4 * i + 1] != 0.0, err);
  from_sweep<NV, MHD>(a.axis, f, fl);
  from_sweep<NV, MHD>(a.axis, ps, pl);
#pragma unroll
  for (int v = 0; v < NV; v++) {
    a.F[(long)i * NV + v] = fl[v];
    a.Pstar[(long)i * NV + v] = pl[v];
  }
  if (err) atomicOr(a.errword, err);
}

#include "stage_helpers.h"
#include "rows_tiling.h"
#include "stage_rows2.h"

// ---------------------------------------------------------------------------
// cooling source term, one thread per on-grid cell of the planes a stage launch updates:
// calc_noRT_microphysics_dU (time_integrator.cpp:438-489) -> mp_only_cooling::TimeUpdateMP.  Only the
// energy component of dU changes, so the kernel leaves dE = PtoU(p_new)[ERG] - PtoU(P)[ERG] (the same two
// PtoU evaluations the reference subtracts) for k_stage_rows2 to start its dU from.  The adaptive
// Cash-Karp loop diverges from cell to cell: here it runs at full occupancy, outside the stage kernel.
// ---------------------------------------------------------------------------
template <int EQ, int NTR>
__global__ __launch_bounds__(256) void k_cooling_dE(const StageArgs a)
{
  typedef Eqn<EQ, NTR> E;
  constexpr int NV = E::NV;
  const unsigned gx = (a.g.ng[0] + 63) / 64, gy = (a.g.ng[1] + 3) / 4;
  const unsigned np1 = (unsigned)(a.kz1 - a.kz0), np2 = (a.kz3 > a.kz2) ? (unsigned)(a.kz3 - a.kz2) : 0u;
  const unsigned ntile = gx * gy * (np1 + np2);
  // The rate tables (temperature grid, five rates and their slopes: 11 NT doubles, 17.6 KB at NT = 200) go to LDS
  // first: an Edot is a bisection (8 dependent look-ups) plus 10 table reads, a Cash-Karp step six of those, and a
  // thread is bound by the latency of that chain -- LDS answers several times sooner than the L1 / L2 path.
  __shared__ double ctab[11 * PION_COOL_NT_MAX];
  const int NT = a.cool.NT;
  for (int i = threadIdx.x; i < NT; i += 256) ctab[i] = a.cool.T[i];
  for (int i = threadIdx.x; i < 5 * NT; i += 256) {
    ctab[NT + i] = a.cool.tab[i];
    ctab[6 * NT + i] = a.cool.slope[i];
  }
  __syncthreads();
  CoolDev cool = a.cool;
  cool.T = ctab;
  cool.tab = ctab + NT;
  cool.slope = ctab + 6 * NT;
  const unsigned t = (unsigned)xcd_tile(blockIdx.x, ntile);
  if (t >= ntile) return;
  const int ix = (int)((t % gx) * 64 + (threadIdx.x & 63));
  const int iy = (int)(((t / gx) % gy) * 4 + (threadIdx.x >> 6));
  const unsigned pz = t / (gx * gy);
  const int iz = (pz < np1) ? a.kz0 + (int)pz : a.kz2 + (int)(pz - np1);
  if (ix >= a.g.ng[0] || iy >= a.g.ng[1]) return;
  const long nc = a.g.ncell;
  const long c = (long)(ix + a.g.nbc[0]) + a.g.sy * (iy + a.g.nbc[1]) + a.g.sz * (iz + a.g.nbc[2]);
  double dE = 0.0;
  if (a.flags[c] & 4 /*ISDOMAIN*/) {
    const double g = a.fc.gamma;
    int err = 0;
#ifdef PION_FAST_MATH
    // fast build: only the pressure changes, so PtoU(p_new)[ERG] - PtoU(P)[ERG] = (p_new - p) / (gamma - 1) -- the
    // kinetic (and magnetic) energy the reference adds to both terms and subtracts again is not read at all (two
    // variables instead of NV per cell; the difference to the reference form is the rounding of that cancellation)
    const double ro = a.Pc[qRO * nc + c], pg = a.Pc[qPG * nc + c];
    const double pnew = Cooling::time_update(cool, ro, pg, a.dt, g, err);
    dE = (pnew - pg) / (g - 1.0);
#else
    double P0[NV], pn[NV], ui[NV], uf[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) pn[v] = P0[v] = a.Pc[v * nc + c];
    pn[qPG] = Cooling::time_update(cool, P0[qRO], P0[qPG], a.dt, g, err);
    E::PtoU(P0, ui, g);
    E::PtoU(pn, uf, g);
    dE = uf[uERG] - ui[uERG];
#endif
    if (err) atomicOr(a.errword, err);
  }
  a.dE[c] = dE;
}
template <int EQ, int NTR>
static int cooling_go(const StageArgs &a, hipStream_t s)
{
  const unsigned gx = (a.g.ng[0] + 63) / 64, gy = (a.g.ng[1] + 3) / 4;
  const unsigned np = (unsigned)(a.kz1 - a.kz0) + ((a.kz3 > a.kz2) ? (unsigned)(a.kz3 - a.kz2) : 0u);
  const unsigned ntile = gx * gy * np;
  hipLaunchKernelGGL((k_cooling_dE<EQ, NTR>), dim3(((ntile + 7) / 8) * 8), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

template <int EQ, int NTR, int SOLVER>
static int stage_go(const StageArgs &a, hipStream_t s)
{
  if (a.use_march != 0 && ((a.g.ndim == 3 && a.g.nbc[2] >= 2) || (a.g.ndim == 2 && a.g.cyl != 2 && a.g.nbc[1] >= 2)))
    return stage_rows2_go<EQ, NTR, SOLVER>(a, s);
  const int nbx = (a.g.ng[0] + 63) / 64, nby = (a.g.ng[1] + 3) / 4;
  const long ntiles = (long)nbx * nby * a.g.ng[2];
  const long nblocks = ((ntiles + 7) / 8) * 8;
  hipLaunchKernelGGL((k_stage<EQ, NTR, SOLVER>), dim3((unsigned)nblocks), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}
template <int EQ, int NTR, int SOLVER>
static int flux_go(const FluxTestArgs &a, hipStream_t s)
{
  hipLaunchKernelGGL((k_flux_test<EQ, NTR, SOLVER>), dim3((a.n + 255) / 256), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

#define PION_SOLVER_CASES_HD(FN, EQ, NTR, A, S)          \
  switch (A.solver) {                                    \
    case 0: return FN<EQ, NTR, 0>(A, S);                 \
    case 1: return FN<EQ, NTR, 1>(A, S);                 \
    case 2: return FN<EQ, NTR, 2>(A, S);                 \
    case 3: return FN<EQ, NTR, 3>(A, S);                 \
    case 4: return FN<EQ, NTR, 4>(A, S);                 \
    case 5: return FN<EQ, NTR, 5>(A, S);                 \
    case 6: return FN<EQ, NTR, 6>(A, S);                 \
    case 8: return FN<EQ, NTR, 8>(A, S);                 \
    default: return -1;                                  \
  }
#define PION_SOLVER_CASES_MHD(FN, EQ, NTR, A, S)         \
  switch (A.solver) {                                    \
    case 0: return FN<EQ, NTR, 0>(A, S);                 \
    case 1: return FN<EQ, NTR, 1>(A, S);                 \
    case 4: return FN<EQ, NTR, 4>(A, S);                 \
    case 7: return FN<EQ, NTR, 7>(A, S);                 \
    case 8: return FN<EQ, NTR, 8>(A, S);                 \
    default: return -1;                                  \
  }

#if PION_EQSEL == 1
#define PION_EQ EQEUL
#define PION_CASES PION_SOLVER_CASES_HD
#define PION_SUFFIX hd
#elif PION_EQSEL == 2
#define PION_EQ EQMHD
#define PION_CASES PION_SOLVER_CASES_MHD
#define PION_SUFFIX mhd
#else
#define PION_EQ EQGLM
#define PION_CASES PION_SOLVER_CASES_MHD
#define PION_SUFFIX glm
#endif
#define PION_CAT2(a, b) a##b
#define PION_CAT(a, b) PION_CAT2(a, b)

#ifdef PION_PROBE
// register / ISA probe builds (profiles/tools/probe_regs.sh): instantiate the production instances only
int PION_CAT(launch_stage_, PION_SUFFIX)(const StageArgs &a, hipStream_t s)
{
  return stage_rows2_go<PION_EQ, 0, (PION_EQSEL == 1 ? 4 : 7)>(a, s);
}
#else
int PION_CAT(launch_stage_, PION_SUFFIX)(const StageArgs &a, hipStream_t s)
{
  switch (a.ntracer) {
    case 0: PION_CASES(stage_go, PION_EQ, 0, a, s)
    case 1: PION_CASES(stage_go, PION_EQ, 1, a, s)
    case 2: PION_CASES(stage_go, PION_EQ, 2, a, s)
    default: return -1;
  }
}
int PION_CAT(launch_cooling_dE_, PION_SUFFIX)(const StageArgs &a, hipStream_t s)
{
  switch (a.ntracer) {
    case 0: return cooling_go<PION_EQ, 0>(a, s);
    case 1: return cooling_go<PION_EQ, 1>(a, s);
    case 2: return cooling_go<PION_EQ, 2>(a, s);
    default: return -1;
  }
}
int PION_CAT(launch_flux_test_, PION_SUFFIX)(const FluxTestArgs &a, hipStream_t s)
{
  switch (a.ntracer) {
    case 0: PION_CASES(flux_go, PION_EQ, 0, a, s)
    case 1: PION_CASES(flux_go, PION_EQ, 1, a, s)
    case 2: PION_CASES(flux_go, PION_EQ, 2, a, s)
    default: return -1;
  }
}
#endif  // PION_PROBE

#else  // PION_EQSEL == 0 : shared kernels -----------------------------------

int launch_stage_hd(const StageArgs &a, hipStream_t s);
int launch_stage_mhd(const StageArgs &a, hipStream_t s);
int launch_stage_glm(const StageArgs &a, hipStream_t s);
int launch_cooling_dE_hd(const StageArgs &a, hipStream_t s);
int launch_cooling_dE_mhd(const StageArgs &a, hipStream_t s);
int launch_cooling_dE_glm(const StageArgs &a, hipStream_t s);
int launch_flux_test_hd(const FluxTestArgs &a, hipStream_t s);
int launch_flux_test_mhd(const FluxTestArgs &a, hipStream_t s);
int launch_flux_test_glm(const FluxTestArgs &a, hipStream_t s);

int launch_stage(const StageArgs &a, hipStream_t s)
{
  if (a.eqntype == EQEUL) return launch_stage_hd(a, s);
  if (a.eqntype == EQMHD) return launch_stage_mhd(a, s);
  if (a.eqntype == EQGLM) return launch_stage_glm(a, s);
  return -1;
}
int launch_cooling_dE(const StageArgs &a, hipStream_t s)
{
  if (a.eqntype == EQEUL) return launch_cooling_dE_hd(a, s);
  if (a.eqntype == EQMHD) return launch_cooling_dE_mhd(a, s);
  if (a.eqntype == EQGLM) return launch_cooling_dE_glm(a, s);
  return -1;
}
// rows per wavefront k_stage_rows2 will use (LDS budget), for the host's launch cost model
int stage_rows2_rows(int eq, int ntr, int zslope_lds, int want)
{
  const int nv = ((eq == EQEUL) ? 5 : ((eq == EQMHD) ? 8 : 9)) + ntr;
  const int nz = zslope_lds ? 2 * nv : nv;
  int r = (int)(PION_ROWS2_LDS_BYTES / (sizeof(double) * 4 * nz * 64));
  if (r > 8) r = 8;
  if (want <= 0) {
    // automatic.  The MHD instances need the whole register file of two wavefronts per SIMD: as many rows as two
    // workgroups' LDS allows (fewer Riemann solves per cell).  The Euler instances take ~160 registers, so a
    // THIRD wavefront per SIMD fits if three workgroups' LDS does: rows for 160 KiB / 3 (measured at 512^3,
    // second-order stage with 2 rows instead of 4: Roe-CV 13.8 -> 12.8 ms/step, FVS + tracer + cooling 25.9 -> 23.6)
    if (eq == EQEUL) {
      int r3 = (int)((160 * 1024 / 3) / (sizeof(double) * 4 * nz * 64));
      if (r3 < 1) r3 = 1;
      if (r3 < r) r = r3;
    }
  }
  else if (want < r) r = want;
  return r < 1 ? 1 : r;
}
int launch_flux_test(const FluxTestArgs &a, hipStream_t s)
{
  if (a.eqntype == EQEUL) return launch_flux_test_hd(a, s);
  if (a.eqntype == EQMHD) return launch_flux_test_mhd(a, s);
  if (a.eqntype == EQGLM) return launch_flux_test_glm(a, s);
  return -1;
}
const char *stage_kernel_name(int, int, int) { return "k_stage"; }

// ---------------------------------------------------------------------------
// pre-pass: HLLD switch and H-correction.  One thread per cell INCLUDING ghosts
// (the reference loops FirstPt_All..NextPt_All, solver_eqn_base.cpp:400-412).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prepass_hlld(const PrepassArgs a)
{
  // 3-D launch (64 x 4 threads; grid = x tiles, y tiles, planes of [c0,c1)): the cell coordinates come
  // from the block and thread indices -- the flat-index form spent more on 64-bit div/mod than on physics
  // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2: give every XCD a contiguous
  // range of (x,y,z) tiles (z slowest) so that the y and z neighbours it reads are lines its own L2 holds
  const long nc = a.g.ncell;
  const unsigned gx = (a.g.nga[0] + 63) / 64, gy = (a.g.nga[1] + 3) / 4;
  const long plane = (long)a.g.nga[0] * a.g.nga[1];
  const unsigned npl1 = (unsigned)((a.c1 - a.c0) / plane);
  const unsigned npl = npl1 + ((a.c3 > a.c2) ? (unsigned)((a.c3 - a.c2) / plane) : 0u);
  const unsigned ntile = gx * gy * npl;
  const unsigned t = (unsigned)xcd_tile(blockIdx.x, ntile);
  if (t >= ntile) return;
  int i[3];
  i[0] = (int)((t % gx) * 64 + (threadIdx.x & 63));
  i[1] = (int)(((t / gx) % gy) * 4 + (threadIdx.x >> 6));
  const unsigned pz = t / (gx * gy);
  i[2] = (pz < npl1) ? (int)(a.c0 / plane) + (int)pz : (int)(a.c2 / plane) + (int)(pz - npl1);
  if (i[0] >= a.g.nga[0] || i[1] >= a.g.nga[1]) return;
  const long c = (long)i[0] + a.g.sy * i[1] + a.g.sz * i[2];
  const double dx = a.g.dx;
  // The switch is (div v < 0 && grad > 5): the pressure term decides for almost every cell (grad > 5 only
  // in strong shocks), so it is evaluated first and the three velocity arrays are read only where it can
  // matter -- the same flags from a quarter of the memory traffic.
  // ... and the pressure term itself is screened without its divisions: if |dp| <= 1.6 min(p-, p+) on every
  // axis the sum of the (at most three) quotients is <= 4.8 (1 + a few ulp) < 5 and the flag is 0 whatever
  // the exact value; only the other cells (and the debug outputs) pay for the divisions.
  double divv = 0.0, gradp = 0.0;
  double pp3[3], pn3[3];
  bool steep = (a.gradp != nullptr);
  for (int v = 0; v < a.g.ndim; v++) {
    const long st = (v == 0) ? 1 : ((v == 1) ? a.g.sy : a.g.sz);
    const long n = (i[v] > 0) ? c - st : c;
    const long p = (i[v] < a.g.nga[v] - 1) ? c + st : c;
    pp3[v] = a.S[1 * nc + p];
    pn3[v] = a.S[1 * nc + n];
    if (!(fabs(pp3[v] - pn3[v]) <= 1.6 * fmin(pp3[v], pn3[v]))) steep = true;   // (NaN counts as steep)
  }
  if (steep) {
    // GradZone (VectorOps.cpp:322-368) on the pressure
    for (int v = 0; v < a.g.ndim; v++) gradp += fabs(pp3[v] - pn3[v]) / fmin(pp3[v], pn3[v]);
  }
  if (gradp > 5. || a.divv) {
    for (int v = 0; v < a.g.ndim; v++) {
      const long st = (v == 0) ? 1 : ((v == 1) ? a.g.sy : a.g.sz);
      // Divergence (VectorOps.cpp:377-439): missing neighbour -> this cell, one-sided dx
      const long n = (i[v] > 0) ? c - st : c;
      const long p = (i[v] < a.g.nga[v] - 1) ? c + st : c;
      const double ddx = (n == c || p == c) ? dx : 2.0 * dx;
      if (a.g.cyl == 1 && v == 1) {
        // VectorOps_Cyl::Divergence (VectorOps.cpp:891-972): d(R V_R)/(R dR) between the neighbours' centres of mass
        const double rn = cyl_Rcom(cyl_R(a.g, (n == c) ? i[1] : i[1] - 1), dx);
        const double rp = cyl_Rcom(cyl_R(a.g, (p == c) ? i[1] : i[1] + 1), dx);
        divv += 2.0 * (rp * a.S[(2 + v) * nc + p] - rn * a.S[(2 + v) * nc + n]) / (rp * rp - rn * rn);
      }
      else divv += (a.S[(2 + v) * nc + p] - a.S[(2 + v) * nc + n]) / ddx;
    }
  }
  if (a.divv) a.divv[c] = divv;
  if (a.gradp) a.gradp[c] = gradp;
  // solver_eqn_mhd_adi.cpp:171: (DivV<0 && Grad>5) of either cell switches the interface to HLL
  a.hllflag[c] = (divv < 0. && gradp > 5.) ? 1 : 0;
}

// The same flags, one thread per (x,y) column of a chunk of PION_PREPASS_ZC planes (3-D Cartesian grids, whole
// plane ranges): the pressure of planes k-1, k, k+1 is carried in registers, so each plane of p is read once
// (+ 2 halo planes per chunk) instead of three times from beyond L2, and -- as in k_prepass_hlld -- the pressure
// term is screened without divisions and the velocities are read only in steep cells.
#define PION_PREPASS_ZC 16
__global__ __launch_bounds__(256) void k_prepass_hlld_march(const PrepassArgs a)
{
  const long nc = a.g.ncell;
  // x: 62 cells per wavefront; lanes 0 and 63 only supply their neighbours' x-1 / x+1 pressure through wavefront
  // shuffles (no loads for the x neighbours)
  const unsigned gx = (a.g.nga[0] + 61) / 62, gy = (a.g.nga[1] + 3) / 4;
  const long plane = (long)a.g.nga[0] * a.g.nga[1];
  const int kz0 = (int)(a.c0 / plane), kz1 = (int)(a.c1 / plane);
  const unsigned nch = (unsigned)((kz1 - kz0 + PION_PREPASS_ZC - 1) / PION_PREPASS_ZC);
  const unsigned ntile = gx * gy * nch;
  const unsigned t = (unsigned)xcd_tile(blockIdx.x, ntile);
  if (t >= ntile) return;
  const int lane = (int)(threadIdx.x & 63);
  int ix = (int)((t % gx) * 62) - 1 + lane;
  const bool xwriter = (lane >= 1 && lane <= 62 && ix < a.g.nga[0]);
  if (ix < 0) ix = 0;
  if (ix > a.g.nga[0] - 1) ix = a.g.nga[0] - 1;
  const int iy = (int)(((t / gx) % gy) * 4 + (threadIdx.x >> 6));
  const int k0 = kz0 + (int)(t / (gx * gy)) * PION_PREPASS_ZC;
  const int k1 = (k0 + PION_PREPASS_ZC < kz1) ? k0 + PION_PREPASS_ZC : kz1;
  if (iy >= a.g.nga[1]) return;   // (whole wavefront: iy is uniform)
  const long sy = a.g.sy, sz = a.g.sz;
  const double dx = a.g.dx;
  const double *P = a.S + 1 * nc;
  const bool xl = ix > 0, xh = ix < a.g.nga[0] - 1, yl = iy > 0, yh = iy < a.g.nga[1] - 1;
  // Addressing as in k_stage_rows2: uniform base (the pressure array, moved by the scalar unit from plane to
  // plane) + one 32-bit byte offset per lane and neighbour.  A cell on a face of the array takes itself for the
  // missing neighbour (the reference's one-sided difference): its neighbour offset IS its own offset, so the
  // loop has neither selects nor branches for the faces.  (launch_prepass uses this kernel only while 8 ncell
  // fits 32 bits.)
  const unsigned cb = (unsigned)((long)ix + sy * iy);   // cell inside a plane
  const unsigned o0 = cb * 8u;
  const unsigned oym = yl ? o0 - (unsigned)sy * 8u : o0, oyp = yh ? o0 + (unsigned)sy * 8u : o0;
  const char *Pk = reinterpret_cast<const char *>(P) + sz * 8 * k0;   // plane k (uniform)
  char *Hk = reinterpret_cast<char *>(a.hllflag) + sz * k0;
  double p0 = ldu(Pk, o0), pm = (k0 > 0) ? ldu(Pk - sz * 8, o0) : p0;
  for (int k = k0; k < k1; k++, Pk += sz * 8, Hk += sz) {
    const bool zl = k > 0, zh = k < a.g.nga[2] - 1;
    const unsigned q0 = pin_v(o0), qym = pin_v(oym), qyp = pin_v(oyp);
    const double pz = zh ? ldu(Pk + sz * 8, q0) : p0;
    double pn3[3], pp3[3];
    // x neighbours from the neighbouring lanes (a cell on an x face of the array takes itself; the halo lanes'
    // own results are not written)
    const double pl_ = lane_prev(p0), pr_ = lane_next(p0);
    pn3[0] = xl ? pl_ : p0;
    pp3[0] = xh ? pr_ : p0;
    pn3[1] = ldu(Pk, qym);
    pp3[1] = ldu(Pk, qyp);
    pn3[2] = zl ? pm : p0;
    pp3[2] = pz;
    // |pp - pn| <= 1.6 min(pp, pn), as two comparisons (a NaN fails both and counts as steep, as before)
    bool steep = false;
#pragma unroll
    for (int v = 0; v < 3; v++) {
      const double d = fabs(pp3[v] - pn3[v]);
      if (!(d <= 1.6 * pp3[v]) || !(d <= 1.6 * pn3[v])) steep = true;
    }
    uint8_t flag = 0;
    if (steep) {
      double gradp = 0.0, divv = 0.0;
      for (int v = 0; v < 3; v++) gradp += fabs(pp3[v] - pn3[v]) / fmin(pp3[v], pn3[v]);   // GradZone
      if (gradp > 5.) {
        // (rare: the cell's index is only formed here)
        const long c = (long)cb + sz * ((long)k + opaque_zero());
        const bool lo[3] = {xl, yl, zl}, hi[3] = {xh, yh, zh};
        for (int v = 0; v < 3; v++) {
          const long st = (v == 0) ? 1 : ((v == 1) ? sy : sz);
          const long n = lo[v] ? c - st : c, p = hi[v] ? c + st : c;
          const double ddx = (n == c || p == c) ? dx : 2.0 * dx;
          divv += (a.S[(2 + v) * nc + p] - a.S[(2 + v) * nc + n]) / ddx;   // Divergence (VectorOps.cpp:377-439)
        }
        flag = (divv < 0.) ? 1 : 0;
      }
    }
    if (xwriter) stub(Hk, pin_v(cb), flag);
    pm = p0;
    p0 = pz;
  }
}

// eta of interface (c | c+st) along each axis, stored at c (calc_Hcorrection; the last cell of a
// column gets no value and keeps 0; first/last cells of a column have zero slope)
template <int NVMAX>
__global__ __launch_bounds__(256) void k_prepass_hcorr(const PrepassArgs a)
{
  const long c = a.c0 + (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nc = a.g.ncell;
  if (c >= a.c1) return;
  int i[3];
  i[0] = (int)(c % a.g.nga[0]);
  i[1] = (int)((c / a.g.nga[0]) % a.g.nga[1]);
  i[2] = (int)(c / ((long)a.g.nga[0] * a.g.nga[1]));
  const double dx = a.g.dx, g = a.gamma;
  const bool mhd = (a.eqntype != EQEUL);
  const bool oa2 = (a.space_ooa == 2);
  for (int ax = 0; ax < a.g.ndim; ax++) {
    const long st = (ax == 0) ? 1 : ((ax == 1) ? a.g.sy : a.g.sz);
    const int n = a.g.nga[ax], k = i[ax];
    double eta = 0.0;
    if (k < n - 1) {
      // sweep-frame variables needed by maxspeed and v_n: rho, p, v_n, (B_n, B_t1, B_t2)
      double eL[8], eR[8];
      const int nvs = mhd ? 8 : 5;
      for (int s = 0; s < 8; s++) {
        if (s >= nvs) {
          eL[s] = eR[s] = 0.0;
          continue;
        }
        const double *b = a.S + (long)(mhd ? rotvar<true>(ax, s) : rotvar<false>(ax, s)) * nc + c;
        const double q0 = b[0], q1 = b[st];
        double s0 = 0.0, s1 = 0.0;
        if (oa2 && a.g.cyl == 2) {
          // VectorOps_Sph::SetSlope / SetEdgeState (VectorOps_spherical.cpp:294-440); k = all-cell R index
          const double R0 = sph_R(a.g, k), R1 = sph_R(a.g, k + 1);
          const double c0 = sph_Rcom(R0, dx), c1 = sph_Rcom(R1, dx);
          if (k > 0) s0 = avg_falle((q0 - b[-st]) / (c0 - sph_Rcom(sph_R(a.g, k - 1), dx)), (q1 - q0) / (c1 - c0));
          if (k + 1 < n - 1) s1 = avg_falle((q1 - q0) / (c1 - c0), (b[2 * st] - q1) / (sph_Rcom(sph_R(a.g, k + 2), dx) - c1));
          eL[s] = q0 + s0 * (R0 + 0.5 * dx - c0);
          eR[s] = q1 + s1 * (R1 - 0.5 * dx - c1);
        }
        else if (oa2 && a.g.cyl == 1 && ax == 1) {
          // VectorOps_Cyl::SetSlope / SetEdgeState along R (VectorOps.cpp:1052-1204); k = all-cell R index
          const double R0 = cyl_R(a.g, k), R1 = cyl_R(a.g, k + 1);
          const double c0 = cyl_Rcom(R0, dx), c1 = cyl_Rcom(R1, dx);
          if (k > 0) s0 = avg_falle((q0 - b[-st]) / (c0 - cyl_Rcom(cyl_R(a.g, k - 1), dx)), (q1 - q0) / (c1 - c0));
          if (k + 1 < n - 1) s1 = avg_falle((q1 - q0) / (c1 - c0), (b[2 * st] - q1) / (cyl_Rcom(cyl_R(a.g, k + 2), dx) - c1));
          eL[s] = q0 + s0 * (R0 + dx * 0.5 - c0);
          eR[s] = q1 + s1 * (R1 - dx * 0.5 - c1);
        }
        else if (oa2) {
          if (k > 0) s0 = avg_falle((q0 - b[-st]) / dx, (q1 - q0) / dx);   // first cell: zero slope
          if (k + 1 < n - 1) s1 = avg_falle((q1 - q0) / dx, (b[2 * st] - q1) / dx);  // last cell: zero slope
          eL[s] = q0 + s0 * dx * 0.5;
          eR[s] = q1 - s1 * dx * 0.5;
        }
        else {
          eL[s] = q0;
          eR[s] = q1;
        }
      }
      // set_Hcorrection (solver_eqn_base.cpp:579-599)
      double msL, msR;
      if (mhd) {
        msL = Eqn<EQMHD, 0>::cfast(eL, g);
        msR = Eqn<EQMHD, 0>::cfast(eR, g);
      }
      else {
        msL = Eqn<EQEUL, 0>::chydro(eL, g);
        msR = Eqn<EQEUL, 0>::chydro(eR, g);
      }
      eta = 0.5 * (fabs(eR[qVN] - eL[qVN]) + fabs(msR - msL));
    }
    a.eta[ax * nc + c] = eta;
  }
}

int launch_prepass(const PrepassArgs &a, hipStream_t s)
{
  const unsigned nb = (unsigned)((a.c1 - a.c0 + 255) / 256);
  if (a.hllflag) {
    // [c0,c1) is a whole number of planes (pion_gpu.hip)
    const long plane = (long)a.g.nga[0] * a.g.nga[1];
    const unsigned npl = (unsigned)((a.c1 - a.c0) / plane) + ((a.c3 > a.c2) ? (unsigned)((a.c3 - a.c2) / plane) : 0u);
    const unsigned ntile = (unsigned)((a.g.nga[0] + 63) / 64) * ((a.g.nga[1] + 3) / 4) * npl;
    if (a.g.ndim == 3 && a.g.cyl == 0 && !a.divv && !a.gradp && a.c3 <= a.c2 && npl >= PION_PREPASS_ZC
        && a.g.ncell * 8L < (1L << 32)) {
      const unsigned nch = (npl + PION_PREPASS_ZC - 1) / PION_PREPASS_ZC;
      const unsigned nt = (unsigned)((a.g.nga[0] + 61) / 62) * ((a.g.nga[1] + 3) / 4) * nch;   // 62 cells per wavefront along x
      hipLaunchKernelGGL(k_prepass_hlld_march, dim3(((nt + 7) / 8) * 8), dim3(256), 0, s, a);
    }
    else hipLaunchKernelGGL(k_prepass_hlld, dim3(((ntile + 7) / 8) * 8), dim3(256), 0, s, a);
  }
  if (a.eta) hipLaunchKernelGGL((k_prepass_hcorr<8>), dim3(nb), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// time-step reduction
// ---------------------------------------------------------------------------
PDEV double wave_min(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(v, off, 64);
    v = (o < v) ? o : v;
  }
  return v;
}

__global__ __launch_bounds__(256) void k_dt(const DtArgs a)
{
  const long nc = a.g.ncell;
  const long non = (long)a.g.ng[0] * a.g.ng[1] * a.g.ng[2];
  double tdyn = 1.e100, tmp = 1.0e99;
  int err = 0;
  const bool mhd = (a.eqntype != EQEUL);
  const double g = a.gamma;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < non; k += (long)gridDim.x * blockDim.x) {
    const int ix = (int)(k % a.g.ng[0]), iy = (int)((k / a.g.ng[0]) % a.g.ng[1]),
              iz = (int)(k / ((long)a.g.ng[0] * a.g.ng[1]));
    const long c = (long)(ix + a.g.nbc[0]) + a.g.sy * (iy + a.g.nbc[1]) + a.g.sz * (iz + a.g.nbc[2]);
    const uint8_t fl = a.flags[c];
    if ((fl & 8 /*TIMESTEP*/) && !(fl & 2 /*ISBD*/)) {
      // CellTimeStep: solver_eqn_hydro_adi.cpp:460-502 / solver_eqn_mhd_adi.cpp:516-582
      double p[8];
      const int nvs = mhd ? 8 : 5;
      for (int v = 0; v < 8; v++) p[v] = (v < nvs) ? a.P[v * nc + c] : 0.0;
      // (the function the stage kernel's fused reduction calls: a restart, which comes through here, continues bit
      // for bit)
      const double t = mhd ? cell_dt<EQMHD>(p, a.g.ndim, g, a.g.dx, a.cfl) : cell_dt<EQEUL>(p, a.g.ndim, g, a.g.dx, a.cfl);
      if (!(t > 0.0)) err |= ERR_BAD_DT;
      tdyn = (t < tdyn) ? t : tdyn;
    }
    if (a.do_mp && !(fl & 2) && (fl & 16)) {
      const double t = Cooling::timescale(a.cool, a.Ph[0 * nc + c], a.Ph[1 * nc + c], g);
      tmp = (t < tmp) ? t : tmp;
    }
  }
  tdyn = wave_min(tdyn);
  tmp = wave_min(tmp);
  if ((threadIdx.x & 63) == 0) {
    // positive doubles order like their bit patterns
    atomicMin(&a.result[0], (unsigned long long)__double_as_longlong(tdyn));
    atomicMin(&a.result[1], (unsigned long long)__double_as_longlong(tmp));
  }
  if (err) atomicOr(a.errword, err);
}

// The cooling time alone (calc_microphysics_dt -> get_mp_timescales_no_radiation, calc_timestep.cpp:405-507;
// mp_only_cooling::timescales, mp_only_cooling.cpp:333-368), rate tables in LDS as in k_cooling_dE: a time scale is two
// Edot evaluations = two bisections of eight dependent look-ups each, which through global memory cost the fused
// reduction of the stage kernel a third of its run time (Wind3D 256^3, second-order instance: 1.58 ms with it, 1.03
// without; this kernel: see DESIGN.md s4.1).  Min over cells with !isbd && isleaf into result[1].
__global__ __launch_bounds__(256) void k_dt_mp(const DtArgs a)
{
  __shared__ double ctab[11 * PION_COOL_NT_MAX];
  const int NT = a.cool.NT;
  for (int i = threadIdx.x; i < NT; i += 256) ctab[i] = a.cool.T[i];
  for (int i = threadIdx.x; i < 5 * NT; i += 256) {
    ctab[NT + i] = a.cool.tab[i];
    ctab[6 * NT + i] = a.cool.slope[i];
  }
  __syncthreads();
  CoolDev cool = a.cool;
  cool.T = ctab;
  cool.tab = ctab + NT;
  cool.slope = ctab + 6 * NT;
  const long nc = a.g.ncell;
  const long non = (long)a.g.ng[0] * a.g.ng[1] * a.g.ng[2];
  double tmp = 1.0e99;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < non; k += (long)gridDim.x * blockDim.x) {
    const int ix = (int)(k % a.g.ng[0]), iy = (int)((k / a.g.ng[0]) % a.g.ng[1]),
              iz = (int)(k / ((long)a.g.ng[0] * a.g.ng[1]));
    const long c = (long)(ix + a.g.nbc[0]) + a.g.sy * (iy + a.g.nbc[1]) + a.g.sz * (iz + a.g.nbc[2]);
    const uint8_t fl = a.flags[c];
    if (!(fl & 2) && (fl & 16)) {
      const double t = Cooling::timescale(cool, a.Ph[0 * nc + c], a.Ph[1 * nc + c], a.gamma);
      tmp = (t < tmp) ? t : tmp;
    }
  }
  tmp = wave_min(tmp);
  if ((threadIdx.x & 63) == 0) atomicMin(&a.result[1], (unsigned long long)__double_as_longlong(tmp));
}
int launch_dt_mp(const DtArgs &a, hipStream_t s)
{
  const long non = (long)a.g.ng[0] * a.g.ng[1] * a.g.ng[2];
  long nb = (non + 255) / 256;
  if (nb > 2048) nb = 2048;   // grid-stride, eight blocks per CU: the tables are staged once per block
  hipLaunchKernelGGL(k_dt_mp, dim3((unsigned)nb), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

int launch_dt(const DtArgs &a, hipStream_t s)
{
  const long non = (long)a.g.ng[0] * a.g.ng[1] * a.g.ng[2];
  long nb = (non + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(k_dt, dim3((unsigned)nb), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// cooling test seams
// ---------------------------------------------------------------------------
__global__ void k_cool_update(const CoolTestArgs a)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  int err = 0;
  const double *p = a.Pin + (long)i * a.nvar;
  double *o = a.Pout + (long)i * a.nvar;
  for (int v = 0; v < a.nvar; v++) o[v] = p[v];
  o[1] = Cooling::time_update(a.cool, p[0], p[1], a.dt, a.gamma, err);
  if (err) atomicOr(a.errword, err);
}
__global__ void k_cool_edot(const CoolTestArgs a)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  a.edot[i] = Cooling::edot(a.cool, a.rho[i], a.T[i]);
}
// mp_only_cooling::timescales (cooling time only) of n independent cells: edot[i] = t_cool
__global__ void k_cool_timescale(const CoolTestArgs a)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const double *p = a.Pin + (long)i * a.nvar;
  a.edot[i] = Cooling::timescale(a.cool, p[0], p[1], a.gamma);
}
int launch_cool_timescale(const CoolTestArgs &a, hipStream_t s)
{
  hipLaunchKernelGGL(k_cool_timescale, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}
int launch_cool_update(const CoolTestArgs &a, hipStream_t s)
{
  hipLaunchKernelGGL(k_cool_update, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}
int launch_cool_edot(const CoolTestArgs &a, hipStream_t s)
{
  hipLaunchKernelGGL(k_cool_edot, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}
#endif  // PION_EQSEL

}  // namespace PION_FPNS
}  // namespace pion
