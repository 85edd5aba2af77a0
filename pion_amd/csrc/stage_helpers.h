// stage_helpers.h -- helpers shared by the 3-D stage kernel (stage_rows2.h): sweep-frame loads, slopes, the
// per-axis source + flux-difference update, CellAdvanceTime, CellTimeStep, the wavefront minimum.
//
// Included by kernels_fp.hip inside namespace pion::PION_FPNS.  (The round-1 kernels k_stage_march -- one
// row per wavefront, z carry in registers, 4 Riemann solves per cell -- and k_stage_rows -- R = 4 rows per
// wavefront at one wavefront per SIMD, 500 registers -- lived in this file (then stage_march.h) and in stage_rows.h (now rows_tiling.h); k_stage_rows2
// superseded both, DESIGN.md s4 keeps their measurements.)
#ifndef PION_STAGE_HELPERS_H
#define PION_STAGE_HELPERS_H

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
  if (no_mp) mp.present = false;  // compile-time constant in the PLAIN instances of k_stage_rows2
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


#endif
