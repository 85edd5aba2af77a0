// stage_march.h -- the 3-D production stage kernel: one wavefront per x-pencil, marching along z.
//
// Included by kernels_fp.hip inside namespace pion::PION_FPNS.
//
// Work decomposition (replaces the per-column pointer walk of
// time_integrator::dynamics_dU_column, sim_control/time_integrator.cpp:645-873):
//   * a wavefront owns 64 consecutive cells of one x-row, lanes 1..62 produce output (the two
//     end lanes only supply their neighbour's interface data), and marches that row along z
//     over a chunk of planes;
//   * x direction: every lane reconstructs its own cell once (one minmod slope, two edge
//     states) and evaluates ONE Riemann problem, at its + face; the right edge state comes from
//     lane+1 and the - face flux from lane-1 by wavefront shuffles;
//   * z direction: the slope of the current plane and the flux through the lower z face are
//     carried in registers from the previous plane, so each plane costs one slope and one
//     Riemann problem;
//   * y direction: both faces are rebuilt by the lane itself (3 slopes, 2 Riemann problems);
//   * all Riemann problems of a plane go through ONE inlined copy of the flux code (a uniform
//     4-iteration task loop), which keeps the kernel inside the instruction cache.
// Per cell and stage: 5 slopes + 4 Riemann solves instead of the 9 + 6 of the cell-per-thread
// kernel.  The contributions are applied to dU in the reference's order (cooling, x, y, z;
// per axis Powell/GLM of the lower face, of the upper face, flux difference), so the strict
// build stays bit-identical.
#ifndef PION_STAGE_MARCH_H
#define PION_STAGE_MARCH_H

template <int NV, bool MHD>
PDEV void load_rot(const double *S, const long nc, const int ax, const long c, double *q)
{
#pragma unroll
  for (int s = 0; s < NV; s++) q[s] = S[(long)rotvar<MHD>(ax, s) * nc + c];
}

// SetSlope for one cell from its two neighbours (VectorOps.cpp:578-617)
template <int NV>
PDEV void slope3(const double *qm, const double *q0, const double *qp, const double dx, const bool oa2, double *s)
{
#pragma unroll
  for (int v = 0; v < NV; v++) s[v] = oa2 ? avg_falle((q0[v] - qm[v]) / dx, (qp[v] - q0[v]) / dx) : 0.0;
}

// MHDsource of both faces + dU_Cell for one axis, in the sweep frame.  q0 is this cell; bnm/sim
// and bnp/sip are B_n and psi of the lower and upper neighbour cells (cell-centred Ph values).
template <int EQ, int NV>
PDEV void apply_axis(double *d, const double *q0, const double bnm, const double sim, const double bnp,
                     const double sip, const double *Fm, const double *Fp, const double dt, const double dx)
{
#ifdef PION_FAST_MATH
  // fast build: the two face terms of the Powell (and GLM) source share their state factor, and the
  // face means differ by (B_n^- - B_n^+)/2: one fused multiply-add per component (the reference adds
  // the lower-face and subtracts the upper-face term separately; same sum up to rounding)
  const double dtdx = dt / dx;
  if constexpr (EQ != EQEUL) {
    const double uB = q0[qBN] * q0[qVN] + q0[qBT1] * q0[qVT1] + q0[qBT2] * q0[qVT2];
    const double kb = dtdx * (0.5 * (bnm - bnp));
    d[uMN] += kb * q0[qBN];
    d[uMT1] += kb * q0[qBT1];
    d[uMT2] += kb * q0[qBT2];
    d[uERG] += kb * uB;
    d[uBN] += kb * q0[qVN];
    d[uBT1] += kb * q0[qVT1];
    d[uBT2] += kb * q0[qVT2];
    if constexpr (EQ == EQGLM) {
      const double ks = dtdx * (0.5 * (sim - sip)) * q0[qVN];
      d[uERG] += ks * q0[qSI];
      d[uPSI] += ks;
    }
  }
#pragma unroll
  for (int s = 0; s < NV; s++) d[s] += dtdx * (Fm[s] - Fp[s]);
  return;
#endif
  if constexpr (EQ != EQEUL) {
    const double uB = q0[qBN] * q0[qVN] + q0[qBT1] * q0[qVT1] + q0[qBT2] * q0[qVT2];
    const double bm0 = 0.5 * (bnm + q0[qBN]);
    d[uMN] += dt * bm0 * (q0[qBN]) / dx;
    d[uMT1] += dt * bm0 * (q0[qBT1]) / dx;
    d[uMT2] += dt * bm0 * (q0[qBT2]) / dx;
    d[uERG] += dt * bm0 * (uB) / dx;
    d[uBN] += dt * bm0 * (q0[qVN]) / dx;
    d[uBT1] += dt * bm0 * (q0[qVT1]) / dx;
    d[uBT2] += dt * bm0 * (q0[qVT2]) / dx;
    if constexpr (EQ == EQGLM) {
      const double sm0 = 0.5 * (sim + q0[qSI]);
      d[uERG] += dt * sm0 * (q0[qVN] * q0[qSI]) / dx;
      d[uPSI] += dt * sm0 * q0[qVN] / dx;
    }
    const double bm1 = 0.5 * (q0[qBN] + bnp);
    d[uMN] -= dt * bm1 * (q0[qBN]) / dx;
    d[uMT1] -= dt * bm1 * (q0[qBT1]) / dx;
    d[uMT2] -= dt * bm1 * (q0[qBT2]) / dx;
    d[uERG] -= dt * bm1 * (uB) / dx;
    d[uBN] -= dt * bm1 * (q0[qVN]) / dx;
    d[uBT1] -= dt * bm1 * (q0[qVT1]) / dx;
    d[uBT2] -= dt * bm1 * (q0[qVT2]) / dx;
    if constexpr (EQ == EQGLM) {
      const double sm1 = 0.5 * (q0[qSI] + sip);
      d[uERG] -= dt * sm1 * (q0[qVN] * q0[qSI]) / dx;
      d[uPSI] -= dt * sm1 * q0[qVN] / dx;
    }
  }
#pragma unroll
  for (int s = 0; s < NV; s++) {
    const double u1 = (Fm[s] - Fp[s]) / dx;
    d[s] += dt * u1;
  }
}

// CellAdvanceTime + temperature clamp (see k_stage): P0 + dU -> Pf
template <int EQ, int NTR>
PDEV void cell_update(const StageArgs &a, const double *P0, const double *dU, int &err, double *Pf,
                      const bool no_mp = false)
{
  typedef Eqn<EQ, NTR> E;
  constexpr int NV = E::NV;
  const double g = a.fc.gamma;
  double u1[NV];
  MPd mp = a.fc.mp;
  if (no_mp) mp.present = false;  // compile-time constant in the PLAIN instances of k_stage_rows
  if (mp.present) {
    double Pi[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) Pi[v] = P0[v];
    E::apply_sCMA(Pi);
    E::PtoU(Pi, u1, g);
  }
  else E::PtoU(P0, u1, g);
#pragma unroll
  for (int v = 0; v < NV; v++) u1[v] += dU[v];
  E::UtoP(u1, Pf, a.fc.min_temp, g, mp, err);
  if (mp.present) E::apply_sCMA(Pf);
  if constexpr (EQ == EQGLM) Pf[qSI] *= a.glm_damp;
  if (mp.present) {
    const double T = Pf[qPG] * mp.Mu_tot_over_kB / Pf[qRO];
    if (T > a.max_temp) Pf[qPG] = Pf[qRO] * a.max_temp / mp.Mu_tot_over_kB;
  }
}
// ... and store
template <int EQ, int NTR>
PDEV void cell_update_store(const StageArgs &a, const long c, const double *P0, const double *dU, int &err,
                           double *Pfout = nullptr, const bool no_mp = false)
{
  constexpr int NV = Eqn<EQ, NTR>::NV;
  const long nc = a.g.ncell;
  double Pf[NV];
  cell_update<EQ, NTR>(a, P0, dU, err, Pf, no_mp);
#pragma unroll
  for (int v = 0; v < NV; v++) a.out[v * nc + c] = Pf[v];
  if (Pfout) {
#pragma unroll
    for (int v = 0; v < NV; v++) Pfout[v] = Pf[v];
  }
}

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
    temp = 0.0;
    for (int v = 0; v < ndim; v++) temp += p[2 + v] * p[2 + v];
    temp = sqrt(temp);
    temp += Eqn<EQEUL, 0>::chydro(p, g);
  }
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

PDEV double wave_min64(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double w = __shfl_xor(v, o, 64);
    v = (w < v) ? w : v;
  }
  return v;
}

#define PION_MARCH_XT 62  // output cells per wavefront along x

template <int EQ, int NTR, int SOLVER>
__global__ __launch_bounds__(256) void k_stage_march(const StageArgs a)
{
  typedef Eqn<EQ, NTR> E;
  typedef Flux<EQ, NTR, SOLVER> FX;
  constexpr int NV = E::NV;
  constexpr bool MHD = E::MHD;

  const int ntx = (a.g.ng[0] + PION_MARCH_XT - 1) / PION_MARCH_XT;
  const int nzc = (a.g.ng[2] + a.zchunk - 1) / a.zchunk;
  const long ntiles = (long)ntx * a.g.ng[1] * nzc;
  // four independent wavefronts per workgroup, each with its own pencil
  const long tile = xcd_tile(blockIdx.x, (ntiles + 3) / 4) * 4 + (threadIdx.x >> 6);
  if (tile >= ntiles) return;  // whole wavefront leaves together
  const int tx = (int)(tile % ntx), iy = (int)((tile / ntx) % a.g.ng[1]), cz = (int)(tile / ((long)ntx * a.g.ng[1]));
  const int lane = threadIdx.x & 63;
  int ix = tx * PION_MARCH_XT - 1 + lane;
  const bool writer = (lane >= 1 && lane <= PION_MARCH_XT && ix < a.g.ng[0]);
  if (ix > a.g.ng[0]) ix = a.g.ng[0];  // clamp into the ghost layer; such lanes produce nothing
  const int k0 = cz * a.zchunk;
  const int k1 = (k0 + a.zchunk < a.g.ng[2]) ? k0 + a.zchunk : a.g.ng[2];

  const long nc = a.g.ncell, sy = a.g.sy, sz = a.g.sz;
  const double g = a.fc.gamma, dx = a.g.dx, dt = a.dt;
  const bool oa2 = (a.space_ooa == 2);
  const bool hcorr = (a.fc.artvisc == AV_HCORRECTION || a.fc.artvisc == AV_HCORR_FKJ98);
  int err = 0;

  // cell (ix, iy, k0-1): the priming plane
  long c = (long)(ix + a.g.nbc[0]) + sy * (iy + a.g.nbc[1]) + sz * (k0 - 1 + a.g.nbc[2]);

  // carried along z (z sweep frame): slope of the current plane, flux through its lower face
  double szc[NV], Fz[NV];
  {
    double qa[NV], qb[NV], qc[NV];
    load_rot<NV, MHD>(a.S, nc, 2, c - sz, qa);
    load_rot<NV, MHD>(a.S, nc, 2, c, qb);
    load_rot<NV, MHD>(a.S, nc, 2, c + sz, qc);
    slope3<NV>(qa, qb, qc, dx, oa2, szc);
#pragma unroll
    for (int v = 0; v < NV; v++) Fz[v] = 0.0;
  }

#pragma unroll 1
  for (int k = k0 - 1; k < k1; k++, c += sz) {
    const bool prime = (k == k0 - 1);  // uniform: only the z task runs, nothing is written
    double q0[NV];                     // this cell, lab frame (= x sweep frame)
#pragma unroll
    for (int v = 0; v < NV; v++) q0[v] = a.S[v * nc + c];

    double dU[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) dU[v] = 0.0;
    double P0[NV];
    uint8_t fl = 0;
    if (!prime) {
#pragma unroll
      for (int v = 0; v < NV; v++) P0[v] = a.Pc[v * nc + c];
      fl = a.flags[c];
      if (a.cooling != 0 && (fl & 4)) {
        // calc_noRT_microphysics_dU (time_integrator.cpp:438-489)
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
    }

    // state kept between the tasks of this plane
    double d[NV];                      // dU in the sweep frame of the axis being processed
    double Fkeep[NV];                  // y: flux of the lower face
    double ys0[NV], ysp[NV], yq0[NV], yqp[NV];  // y: slopes/cells needed again for the upper face
    double bnm = 0.0, sim = 0.0, bnp = 0.0, sip = 0.0;

#pragma unroll 1
    for (int t = prime ? 3 : 0; t < 4; t++) {
      double eL[NV], eR[NV], f[NV], pstar[NV];
      long cl;
      long st;
      int ax;
      if (t == 0) {
        // ---- x: own reconstruction, right state from lane+1 -------------------------------
        ax = 0;
        st = 1;
        cl = c;
        double qm[NV], qp[NV], sx[NV];
#pragma unroll
        for (int v = 0; v < NV; v++) {
          qm[v] = a.S[v * nc + c - 1];
          qp[v] = a.S[v * nc + c + 1];
        }
        slope3<NV>(qm, q0, qp, dx, oa2, sx);
#pragma unroll
        for (int v = 0; v < NV; v++) {
          double em;
          if (oa2) {
            eL[v] = q0[v] + sx[v] * dx * 0.5;
            em = q0[v] - sx[v] * dx * 0.5;
          }
          else {
            eL[v] = q0[v];
            em = q0[v];
          }
          eR[v] = __shfl_down(em, 1, 64);
        }
        if constexpr (MHD) {
          bnm = qm[qBN];
          bnp = qp[qBN];
          if constexpr (EQ == EQGLM) {
            sim = qm[qSI];
            sip = qp[qSI];
          }
        }
      }
      else if (t == 1) {
        // ---- y, lower face (c-sy | c): three slopes, kept for the upper face -----------------
        ax = 1;
        st = sy;
        cl = c - sy;
        double qm2[NV], qm1[NV], qp2[NV], sm1[NV];
        load_rot<NV, MHD>(a.S, nc, 1, c - sy, qm1);
        load_rot<NV, MHD>(a.S, nc, 1, c + sy, yqp);
        to_sweep<NV, MHD>(1, q0, yq0);
        if (oa2) {
          load_rot<NV, MHD>(a.S, nc, 1, c - 2 * sy, qm2);
          load_rot<NV, MHD>(a.S, nc, 1, c + 2 * sy, qp2);
        }
        else {
#pragma unroll
          for (int v = 0; v < NV; v++) qm2[v] = qp2[v] = 0.0;
        }
        slope3<NV>(qm2, qm1, yq0, dx, oa2, sm1);
        slope3<NV>(qm1, yq0, yqp, dx, oa2, ys0);
        slope3<NV>(yq0, yqp, qp2, dx, oa2, ysp);
#pragma unroll
        for (int v = 0; v < NV; v++) {
          if (oa2) {
            eL[v] = qm1[v] + sm1[v] * dx * 0.5;
            eR[v] = yq0[v] - ys0[v] * dx * 0.5;
          }
          else {
            eL[v] = qm1[v];
            eR[v] = yq0[v];
          }
        }
        if constexpr (MHD) {
          bnm = qm1[qBN];
          bnp = yqp[qBN];
          if constexpr (EQ == EQGLM) {
            sim = qm1[qSI];
            sip = yqp[qSI];
          }
        }
      }
      else if (t == 2) {
        // ---- y, upper face (c | c+sy) ----------------------------------------------------------
        ax = 1;
        st = sy;
        cl = c;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          if (oa2) {
            eL[v] = yq0[v] + ys0[v] * dx * 0.5;
            eR[v] = yqp[v] - ysp[v] * dx * 0.5;
          }
          else {
            eL[v] = yq0[v];
            eR[v] = yqp[v];
          }
        }
      }
      else {
        // ---- z, upper face (c | c+sz): slope of the next plane, carried slope of this one ------
        ax = 2;
        st = sz;
        cl = c;
        double qp1[NV], qp2[NV], sn[NV];
        load_rot<NV, MHD>(a.S, nc, 2, c + sz, qp1);
        to_sweep<NV, MHD>(2, q0, yq0);  // yq0 is free again: this cell in the z frame
        if (oa2) load_rot<NV, MHD>(a.S, nc, 2, c + 2 * sz, qp2);
        else {
#pragma unroll
          for (int v = 0; v < NV; v++) qp2[v] = 0.0;
        }
        slope3<NV>(yq0, qp1, qp2, dx, oa2, sn);
#pragma unroll
        for (int v = 0; v < NV; v++) {
          if (oa2) {
            eL[v] = yq0[v] + szc[v] * dx * 0.5;
            eR[v] = qp1[v] - sn[v] * dx * 0.5;
          }
          else {
            eL[v] = yq0[v];
            eR[v] = qp1[v];
          }
          szc[v] = sn[v];  // becomes the current plane's slope in the next iteration
        }
        if constexpr (MHD) {
          bnp = qp1[qBN];
          bnm = a.S[(long)rotvar<MHD>(2, qBN) * nc + c - sz];
          if constexpr (EQ == EQGLM) {
            sip = qp1[qSI];
            sim = a.S[(long)qSI * nc + c - sz];
          }
        }
      }

      // ---- the one inlined Riemann solve ----------------------------------------------------
      double hc_eta = 0.0;
      if (hcorr) hc_eta = select_hcorr_eta(a, ax, cl, st);
      bool use_hll = false;
      if constexpr (MHD && SOLVER == FLUX_RS_HLLD) use_hll = (a.hllflag[cl] | a.hllflag[cl + st]) != 0;
      FX::intercell_flux(eL, eR, f, pstar, a.fc, hc_eta, use_hll, err);

      // ---- apply ------------------------------------------------------------------------------
      if (t == 0) {
        double Fm[NV];
#pragma unroll
        for (int v = 0; v < NV; v++) Fm[v] = __shfl_up(f[v], 1, 64);
#pragma unroll
        for (int v = 0; v < NV; v++) d[v] = dU[v];
        apply_axis<EQ, NV>(d, q0, bnm, sim, bnp, sip, Fm, f, dt, dx);
#pragma unroll
        for (int v = 0; v < NV; v++) dU[v] = d[v];
      }
      else if (t == 1) {
#pragma unroll
        for (int v = 0; v < NV; v++) Fkeep[v] = f[v];
      }
      else if (t == 2) {
        to_sweep<NV, MHD>(1, dU, d);
        apply_axis<EQ, NV>(d, yq0, bnm, sim, bnp, sip, Fkeep, f, dt, dx);
        from_sweep<NV, MHD>(1, d, dU);
      }
      else {
        if (!prime) {
          to_sweep<NV, MHD>(2, dU, d);
          apply_axis<EQ, NV>(d, yq0, bnm, sim, bnp, sip, Fz, f, dt, dx);
          from_sweep<NV, MHD>(2, d, dU);
        }
#pragma unroll
        for (int v = 0; v < NV; v++) Fz[v] = f[v];
      }
    }

    if (!prime && writer) {
      if (!(fl & 4) || !(fl & 16)) {
#pragma unroll
        for (int v = 0; v < NV; v++) a.out[v * nc + c] = P0[v];
      }
      else cell_update_store<EQ, NTR>(a, c, P0, dU, err);
    }
  }
  if (err) atomicOr(a.errword, err);
}

template <int EQ, int NTR, int SOLVER>
static int stage_march_go(const StageArgs &a, hipStream_t s)
{
  const int ntx = (a.g.ng[0] + PION_MARCH_XT - 1) / PION_MARCH_XT;
  const int nzc = (a.g.ng[2] + a.zchunk - 1) / a.zchunk;
  const long ntiles = (long)ntx * a.g.ng[1] * nzc;
  const long nblocks = (((ntiles + 3) / 4 + 7) / 8) * 8;
  hipLaunchKernelGGL((k_stage_march<EQ, NTR, SOLVER>), dim3((unsigned)nblocks), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

#endif
