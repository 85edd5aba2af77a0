// stage_rows2.h -- k_stage_rows2: the 3-D production stage kernel, two wavefronts per SIMD (MHD) or three (Euler).
//
// Included by kernels_fp.hip inside namespace pion::PION_FPNS (after dev_addr.h, stage_helpers.h and rows_tiling.h,
// whose helpers uni / ldu / pin_v / opaque_zero, apply_axis / cell_update, rows_tiling it uses; cell_dt is
// kernels_fp.hip's).
//
// Same decomposition as round 1's k_stage_rows (removed; one wavefront per x-pencil of 64 lanes owning R consecutive
// y-rows, marching along z; x fluxes shared by wavefront shuffles, the y flux and the next row's y slope
// carried from row to row in registers, the z-carried state of every row in LDS) rebuilt around the
// round-1 profile: that kernel needed 500 registers (one wavefront per SIMD), spent 22 % of its wave
// cycles parked on s_waitcnt with nothing else to issue, and 10 % of its instructions copying between
// VGPRs and AGPRs.  Here
//   * the kernel is built for <= 256 registers (__launch_bounds__(256, 2): two workgroups of four
//     wavefronts per CU, two wavefronts per SIMD), so a wavefront waiting on memory has a partner that
//     issues;
//   * nothing is prefetched into registers across a Riemann solve: what is live across the one inlined
//     flux body is this row's state q0, its dU, and the y carry (flux + slope) -- 4 x NV doubles;
//   * LDS per wavefront halves (80 KiB per workgroup): with ZSL the z slope and the lower z flux of every
//     row are carried (R = 2 rows at nvar 9), without it only the flux is carried and the slope of the
//     current plane is rebuilt from plane k-1 (R = 4 rows at nvar 9: 3.25 instead of 3.5 Riemann solves per
//     cell, one more read of each plane, which comes from L2 / the Infinity Cache);
//   * the first-order stage reads the start-of-step state from the stencil array (they are the same
//     array in that stage), the cooling source arrives as one double per cell from k_cooling;
//   * the Euler instances need ~160 registers: the host picks their rows per wavefront so that THREE workgroups'
//     LDS fits a CU (stage_rows2_rows), and three wavefronts per SIMD run.
// The arithmetic and its order are the reference's (the strict build stays bit-identical to the oracle).
#ifndef PION_STAGE_ROWS2_H
#define PION_STAGE_ROWS2_H

#ifndef PION_ROWS2_NT
#define PION_ROWS2_NT 1   // non-temporal stores of the new state / loads of the once-read start-of-step state (25.35-25.43 vs 25.5-25.6 ms/step; 2 = also the z planes: 25.52)
#endif

PDEV void stu(char *ubase, const unsigned off, const double x)
{
  typedef __attribute__((address_space(1))) char *gc;
  typedef __attribute__((address_space(1))) double *gp;
#if PION_ROWS2_NT
  __builtin_nontemporal_store(x, (gp)((gc)uni(ubase) + off));   // written once, read by the next launch
#else
  *(gp)((gc)uni(ubase) + off) = x;
#endif
}
// a value that is read exactly once by the launch (the start-of-step state of the second-order stage)
PDEV double ldu_once(const char *ubase, const unsigned off)
{
  typedef const __attribute__((address_space(1))) char *gc;
  typedef const __attribute__((address_space(1))) double *gp;
#if PION_ROWS2_NT
  return __builtin_nontemporal_load((gp)((gc)uni(ubase) + off));
#else
  return *(gp)((gc)uni(ubase) + off);
#endif
}
// sweep-frame state of the cell at uniform byte shift `sh` from the lane's cell
template <int NV, bool MHD>
PDEV void load_rot2(const char *Sb, const long ncb, const int ax, const long sh, const unsigned off, double *q)
{
#pragma unroll
  for (int s = 0; s < NV; s++) {
#if PION_ROWS2_NT >= 2
    if (ax == 2) q[s] = ldu_once(Sb + (long)rotvar<MHD>(ax, s) * ncb + sh, off);   // z planes: streamed
    else
#endif
      q[s] = ldu(Sb + (long)rotvar<MHD>(ax, s) * ncb + sh, off);
  }
}

// SetSlope x dx for one cell from its two neighbours (VectorOps.cpp:578-617, AvgFalle :37-59): the edge
// states are q -+ hs / 2 (SetEdgeState :535-571).  Strict build: the reference's slope s = minmod(a/dx, b/dx)
// (division form) times dx, so that q + hs * 0.5 is the reference's q + s * dx * 0.5 bit for bit.  Fast
// build: minmod of the raw differences as a median -- no division, no scaling by dx and back (<= 1 ulp from
// the reference form), no product, compare and select.
template <int NV>
PDEV void hslope3(const double *qm, const double *q0, const double *qp, const double dx, const double thr, double *hs)
{
#pragma unroll
  for (int v = 0; v < NV; v++) {
#ifdef PION_FAST_MATH
    // minmod(a, b) = median(a, b, 0) = max(min(a, b), min(max(a, b), 0)): four min / max instructions
    // (the reference's "a b <= 1e-200" also zeroes slopes below 1e-100 dx, which this form keeps)
    const double a = q0[v] - qm[v], b = qp[v] - q0[v];
    hs[v] = fmx(fmn(a, b), fmn(fmx(a, b), 0.0));
#else
    hs[v] = avg_falle((q0[v] - qm[v]) / dx, (qp[v] - q0[v]) / dx) * dx;
#endif
  }
}

// SetSlope along R of a cylindrical grid (VectorOps_Cyl::SetSlope, VectorOps.cpp:1103-1204): the TRUE slope (not
// x dx) from the differences of the cells' centres of mass cm, c0, cp
template <int NV>
PDEV void cyl_slope3(const double *qm, const double *q0, const double *qp, const double cm, const double c0, const double cp,
                     const bool oa2, double *s)
{
#pragma unroll
  for (int v = 0; v < NV; v++) s[v] = oa2 ? avg_falle((q0[v] - qm[v]) / (c0 - cm), (qp[v] - q0[v]) / (cp - c0)) : 0.0;
}
// The R axis of a cylindrical grid for one cell, in the sweep frame: MHD source of both faces, R-weighted flux
// divergence and geometric source -- the statements of k_stage's cylR branch in the same order
// (cyl_FV_solver_mhd_ideal_adi::MHDsource, solver_eqn_mhd_adi.cpp:1081-1092; VectorOps_Cyl::DivStateVectorComponent,
// VectorOps.cpp:1211-1245; geometric_source, solver_eqn_hydro_adi.cpp:560-590, solver_eqn_mhd_adi.cpp:1001-1030,
// 1175-1210).  Rm1 / R0: centres of the lower neighbour and of this cell; s0: this cell's slope along R.
template <int EQ, int NV>
PDEV void apply_axis_cyl(double *d, const double *q0, const double bnm, const double sim, const double bnp, const double sip,
                         const double *Fm, const double *Fp, const double *s0, const double dt, const double dx,
                         const double Rm1, const double R0, const double Rc, const bool oa2, const double chyp)
{
  constexpr bool MHD = (EQ != EQEUL);
#ifdef PION_FAST_MATH
  // fast build: the same sums regrouped -- one factor per face for the Powell source (its state factor is shared by
  // the two faces), one per face for the R-weighted divergence (exact in real arithmetic, differs by rounding)
  {
    const double rn = R0 - dx * 0.5, rp = R0 + dx * 0.5;
    const double idn = 2.0 * dt / (rp * rp - rn * rn);
    const double wn = idn * rn, wp = idn * rp;
    if constexpr (MHD) {
      const double uB = q0[qBN] * q0[qVN] + q0[qBT1] * q0[qVT1] + q0[qBT2] * q0[qVT2];
      const double kb = wn * (0.5 * (bnm + q0[qBN])) - wp * (0.5 * (q0[qBN] + bnp));
      d[uMN] += kb * q0[qBN];
      d[uMT1] += kb * q0[qBT1];
      d[uMT2] += kb * q0[qBT2];
      d[uERG] += kb * uB;
      d[uBN] += kb * q0[qVN];
      d[uBT1] += kb * q0[qVT1];
      d[uBT2] += kb * q0[qVT2];
      if constexpr (EQ == EQGLM) {
        const double ks = (dt / dx) * (0.5 * (sim - sip)) * q0[qVN];
        d[uERG] += ks * q0[qSI];
        d[uPSI] += ks;
      }
    }
#pragma unroll
    for (int s = 0; s < NV; s++) d[s] += wn * Fm[s] - wp * Fp[s];
    const double iR0 = dt / R0;
    if constexpr (!MHD) {
      d[uMN] += iR0 * (oa2 ? (q0[qPG] + (R0 - Rc) * s0[qPG]) : q0[qPG]);
    }
    else {
      const double pm = (q0[qBN] * q0[qBN] + q0[qBT1] * q0[qBT1] + q0[qBT2] * q0[qBT2]) * 0.5;
      if (oa2) {
        d[uMN] += iR0 * (q0[qPG] + pm + (R0 - Rc) * (s0[qPG] + q0[qBN] * s0[qBN] + q0[qBT1] * s0[qBT1] + q0[qBT2] * s0[qBT2]));
        if constexpr (EQ == EQGLM) d[uBN] += iR0 * chyp * (q0[qSI] + (R0 - Rc) * s0[qSI]);
      }
      else {
        d[uMN] += iR0 * (q0[qPG] + pm);
        if constexpr (EQ == EQGLM) d[uBN] += iR0 * chyp * q0[qSI];
      }
    }
    return;
  }
#endif
  if constexpr (MHD) {
    const double uB = q0[qBN] * q0[qVN] + q0[qBT1] * q0[qVT1] + q0[qBT2] * q0[qVT2];
    const double bm0 = 0.5 * (bnm + q0[qBN]);
    {
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
    if constexpr (EQ == EQGLM) {
      const double sm0 = 0.5 * (sim + q0[qSI]);
      d[uERG] += dt * sm0 * (q0[qVN] * q0[qSI]) / dx;
      d[uPSI] += dt * sm0 * q0[qVN] / dx;
    }
    const double bm1 = 0.5 * (q0[qBN] + bnp);
    {
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
    if constexpr (EQ == EQGLM) {
      const double sm1g = 0.5 * (q0[qSI] + sip);
      d[uERG] -= dt * sm1g * (q0[qVN] * q0[qSI]) / dx;
      d[uPSI] -= dt * sm1g * q0[qVN] / dx;
    }
  }
  double u1[NV];
  const double rp = R0 + dx * 0.5;
  const double rn = rp - dx;
#pragma unroll
  for (int s = 0; s < NV; s++) u1[s] = 2.0 * (rn * Fm[s] - rp * Fp[s]) / (rp * rp - rn * rn);
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
      if constexpr (EQ == EQGLM) u1[uBN] += chyp * (q0[qSI] + (R0 - Rc) * s0[qSI]) / R0;
    }
    else {
      u1[uMN] += (q0[qPG] + pm) / R0;
      if constexpr (EQ == EQGLM) u1[uBN] += chyp * q0[qSI] / R0;
    }
  }
#pragma unroll
  for (int s = 0; s < NV; s++) d[s] += dt * u1[s];
}

#ifndef PION_ROWS2_PF
#define PION_ROWS2_PF 1
#endif
#ifndef PION_ROWS2_YWG
#define PION_ROWS2_YWG 1
#endif
// PION_ROWS2_COPIES=1: the x, y and z tasks of a row as three straight-line copies of the task body; 0: one body in
// a uniform task loop (A/B)
#ifndef PION_ROWS2_COPIES
#define PION_ROWS2_COPIES 1
#endif
#ifndef PION_ROWS2_U0
#define PION_ROWS2_U0 2
#endif

// workgroups per CU the register allocation aims at
#ifndef PION_ROWS2_MINWG
#define PION_ROWS2_MINWG(EQ) 2
#endif
// CYL: the 2-D instance for a cylindrical (z,R) grid (g.cyl == 1): the rows are marched along R, whose geometry enters
// the slopes (differences of the cells' centres of mass), the edge states, the flux divergence and the source terms
// exactly as in k_stage (VectorOps_Cyl, cyl_FV_solver_*); the z axis of the grid is the wavefront's x and stays Cartesian.
template <int EQ, int NTR, int SOLVER, int OAMODE, bool PLAIN, bool ZSL, bool CYL = false>
__global__ __launch_bounds__(256, PION_ROWS2_MINWG(EQ)) void k_stage_rows2(const StageArgs a)
{
  typedef Eqn<EQ, NTR> E;
  typedef Flux<EQ, NTR, SOLVER> FX;
  constexpr int NV = E::NV;
  constexpr bool MHD = E::MHD;
  constexpr int NZ = ZSL ? 2 * NV : NV;   // LDS slots per row: [z slope,] lower z flux
  constexpr bool PF = PION_ROWS2_PF && (OAMODE == 1);
  extern __shared__ double lds[];

  const int R = a.rows;
  const RowsTiling tl = rows_tiling(a);
  const int nyg = tl.nyg;
  const int nzc1 = (a.nzb > 0) ? a.nzb : (a.kz1 - a.kz0 + a.zchunk - 1) / a.zchunk;
  const int nzc = nzc1 + (a.kz3 - a.kz2 + a.zchunk - 1) / a.zchunk;   // chunks of both strips
  // the wavefront number is uniform: say so, and the tile / row / plane loops run on the scalar unit
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // Workgroup -> (x-y tile group, plane chunk).  Workgroups go to the XCDs round robin (blockIdx % 8): each XCD
  // takes an eighth of the x-y tiles -- y-adjacent row groups, whose halo rows are each other's own rows, share its
  // L2 -- through ALL plane chunks, in chunk order: with uneven chunks (long first) every XCD ends on the short ones.
  // (Chunk-major over the whole grid would hand the long chunks to some XCDs and the short ones to others.)
  const int nb4 = (tl.per_chunk + 3) / 4, nb8 = (nb4 + 7) / 8;   // workgroups per chunk; per chunk and XCD
  const int lb = (int)(blockIdx.x >> 3);
  const int cz = lb / nb8, bq = (int)(blockIdx.x & 7) * nb8 + lb % nb8;
  const int tt = bq * 4 + wave;
  if (cz >= nzc || bq >= nb4 || tt >= tl.per_chunk) return;  // whole wavefront leaves together (no block-level barrier is used)
  const int lane = threadIdx.x & 63;
  int ix, jg, jg_first;   // jg: this LANE's row group; jg_first: the wavefront's first (uniform)
  bool writer;
  if (tt < tl.nfull) {
#if PION_ROWS2_YWG
    // the four wavefronts of a workgroup take four y-adjacent row groups of the same x tile: the y halo rows
    // of one are the own rows of the next, read at about the same time on the same CU / L2
    const int per4 = 4 * tl.ntx_full, g4 = tt / per4, r4 = tt - g4 * per4;
    const int m = (nyg - 4 * g4 < 4) ? nyg - 4 * g4 : 4;
    const int tx = r4 / m;
    jg = jg_first = 4 * g4 + r4 % m;
#else
    const int tx = tt % tl.ntx_full;
    jg = jg_first = tt / tl.ntx_full;
#endif
    ix = tx * PION_MARCH_XT - 1 + lane;
    writer = (lane >= 1 && lane <= PION_MARCH_XT && ix < a.g.ng[0]);
  }
  else {
    const int seg = lane / (tl.rem + 2), pos = lane % (tl.rem + 2);
    jg_first = (tt - tl.nfull) * tl.spw;
    jg = jg_first + seg;
    ix = tl.ntx_full * PION_MARCH_XT - 1 + pos;
    writer = (seg < tl.spw && jg < nyg && pos >= 1 && pos <= tl.rem);
    if (jg >= nyg) jg = nyg - 1;   // idle lanes redo the last group, in bounds, and write nothing
  }
  if (ix > a.g.ng[0]) ix = a.g.ng[0];
  const int j0 = jg * R;
  const int nrows = (jg_first * R + R <= a.g.ng[1]) ? R : a.g.ng[1] - jg_first * R;
  const int nrows_l = (j0 + R <= a.g.ng[1]) ? R : a.g.ng[1] - j0;
  int k0, k1;
  if (a.nzb > 0 && cz < nzc1) {
    zchunk_bounds(a.kz1 - a.kz0, a.zcmax, cz, &k0, &k1);   // (scalar unit: cz is uniform)
    k0 += a.kz0;
    k1 += a.kz0;
  }
  else {
    k0 = (cz < nzc1) ? a.kz0 + cz * a.zchunk : a.kz2 + (cz - nzc1) * a.zchunk;
    const int kend = (cz < nzc1) ? a.kz1 : a.kz3;
    k1 = (k0 + a.zchunk < kend) ? k0 + a.zchunk : kend;
  }

  const long nc = a.g.ncell, sy = a.g.sy, sz = a.g.sz;
  const long ncb = nc * 8, syb = sy * 8, szb = sz * 8;   // byte strides (uniform)
  const char *const Sb = reinterpret_cast<const char *>(a.S);
  const char *const Hb = reinterpret_cast<const char *>(a.hllflag);
  const double g = a.fc.gamma, dx = a.g.dx, dt = a.dt;
  const bool oa2 = (OAMODE == 0) ? (a.space_ooa == 2) : (OAMODE == 2);
  // 2-D Cartesian grids run the same kernel without its z part: one "plane", a group of R rows per wavefront marched
  // along y with the flux and the slope carried in registers (2 + 1/R Riemann solves per cell), no LDS
  const bool noz = (a.g.ndim == 2);
  const double thr = PION_VERY_TINY_VALUE * dx * dx;   // AvgFalle's zero test on raw differences (fast build)
  const bool hcorr = PLAIN ? false : (a.fc.artvisc == AV_HCORRECTION || a.fc.artvisc == AV_HCORR_FKJ98);
  // first-order stages read their stencil from the start-of-step array (time_integrator.cpp:151-250:
  // Ph == P at the start of a step), so the centre value is the start-of-step state
  const bool same_pc = (OAMODE == 1) ? true : (a.S == a.Pc);
  FluxCtx fc = a.fc;
  if (PLAIN) fc.mp.present = 0;
  int err = 0;
  double tdyn = 1.e100, tmp = 1.0e99;  // running minima for the fused time-step reduction

  const int zbase = wave * R * NZ * 64 + lane;
#define ZS2(r, s) lds[zbase + ((r) * NZ + (s)) * 64]

  const long crow0 = (long)(ix + a.g.nbc[0]) + sy * (j0 + a.g.nbc[1]) + sz * (k0 - 1 + a.g.nbc[2]);

  // z state of the priming plane k0-1 for every row
#pragma unroll 1
  for (int r = 0; r < (noz ? 0 : nrows); r++) {
    if constexpr (ZSL) {
      const long c = crow0 + sy * ((r < nrows_l) ? r : nrows_l - 1);
      const unsigned off = pin_v((unsigned)c * 8u);
      double qa[NV], qb[NV], qc[NV], s[NV];
      load_rot2<NV, MHD>(Sb, ncb, 2, -szb, off, qa);
      load_rot2<NV, MHD>(Sb, ncb, 2, 0, off, qb);
      load_rot2<NV, MHD>(Sb, ncb, 2, szb, off, qc);
      if (oa2) hslope3<NV>(qa, qb, qc, dx, thr, s);
      else {
#pragma unroll
        for (int v = 0; v < NV; v++) s[v] = 0.0;
      }
#pragma unroll
      for (int v = 0; v < NV; v++) ZS2(r, v) = s[v];
    }
#pragma unroll
    for (int v = 0; v < NV; v++) ZS2(r, NZ - NV + v) = 0.0;
  }

  // PF (first-order instances): what the NEXT task needs from memory -- one row of the state, B_n / psi of
  // the lower neighbour, two switch flags -- is requested before the current Riemann solve and consumed
  // after it, so that the task starts without a wait (the second-order instances need two or three rows
  // per task and have no registers left for them)
  double pf[NV], pfb = 0.0, pfs = 0.0;
  unsigned pfh = 0;
  bool pf_valid = false;   // a request has been made (false for the first task a wavefront runs)
#pragma unroll
  for (int v = 0; v < NV; v++) pf[v] = 0.0;

#pragma unroll 1
  for (int k = noz ? k0 : k0 - 1; k < k1; k++) {
    const bool prime = !noz && (k == k0 - 1);
    // carried from row to row inside this plane (y sweep frame): flux through the upper y face of the
    // previous row, y slope of the row being processed
    double Fy[NV], ysn[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) Fy[v] = ysn[v] = 0.0;
    // (CYL) centres of mass of rows j-1 .. j+2, carried from row to row: one R_com (a division) per row instead of four
    double cyc[4] = {0.0, 0.0, 0.0, 0.0};

#pragma unroll 1
    for (int r = 0; r < nrows; r++) {
      const bool row_ok = (r < nrows_l);
      const long c = crow0 + sy * (row_ok ? r : nrows_l - 1) + sz * (k - (k0 - 1));
      const unsigned off_r = (unsigned)c * 8u, offb_r = (unsigned)c;   // the lane's cell: byte offsets of doubles / flags
      const int jy_r = j0 + (row_ok ? r : nrows_l - 1) + a.g.nbc[1];   // (CYL) all-cell y index of the lane's row
      if constexpr (CYL) {
        if (r == 0 || !row_ok) {
          cyc[0] = cyl_Rcom(cyl_R(a.g, jy_r - 1), dx);
          cyc[1] = cyl_Rcom(cyl_R(a.g, jy_r), dx);
          cyc[2] = cyl_Rcom(cyl_R(a.g, jy_r + 1), dx);
        }
        else {
          cyc[0] = cyc[1];
          cyc[1] = cyc[2];
          cyc[2] = cyc[3];
        }
        cyc[3] = cyl_Rcom(cyl_R(a.g, jy_r + 2), dx);
      }
      const double cy_m1 = cyc[0], cy_0 = cyc[1], cy_p1 = cyc[2], cy_p2 = cyc[3];   // (by value into the task)
      // the row visited after this one (next row of the plane, or the first row of the next plane)
      const long cn = (r + 1 < nrows) ? crow0 + sy * ((r + 1 < nrows_l) ? r + 1 : nrows_l - 1) + sz * (k - (k0 - 1))
                                      : (noz ? c : crow0 + sz * (k + 1 - (k0 - 1)));   // (2-D: no next plane)
      const unsigned offn_r = (unsigned)cn * 8u, offnb_r = (unsigned)cn;
      const unsigned doff_n = offn_r - off_r, doffb_n = offnb_r - offb_r;
      double q0[NV], dU[NV];
      if (PF && pf_valid && !prime) {
#pragma unroll
        for (int v = 0; v < NV; v++) q0[v] = pf[v];   // requested before the previous row's last solve
      }
      else {
        const unsigned o = pin_v(off_r);
#pragma unroll
        for (int v = 0; v < NV; v++) q0[v] = ldu(Sb + v * ncb, o);
      }
#pragma unroll
      for (int v = 0; v < NV; v++) dU[v] = 0.0;
#if defined(PION_FAST_MATH) && PION_ROWS2_U0
      // Fast build, second-order stage of a grid without internal-boundary cells: the start-of-step state is
      // requested HERE, with the row's own loads, and enters as dU = PtoU(P0) -- the new conserved state is then
      // ((U0 + dU_x) + dU_y) + dU_z instead of the reference's U0 + ((dU_x + dU_y) + dU_z) (same sum up to
      // rounding; the strict build keeps the reference's order).  At the end of the row, where the reference form
      // needs P0, a load would be waited for with nothing to overlap it.
      // PION_ROWS2_U0 = 2: the loads go out in the x task, AFTER the loads the x task itself waits for (loads return
      // in issue order), straight into dU's registers, and are converted in place after the x solve: a read of HBM
      // that nothing waits for.
      const bool u0 = PLAIN && !same_pc && !prime && a.plain_cells;
#if PION_ROWS2_U0 == 1
      if (u0) {
        const unsigned o = pin_v(off_r);
        const unsigned zu0 = opaque_zero();
        const char *const Pcb0 = reinterpret_cast<const char *>(a.Pc) + zu0;
        double P0[NV];
#pragma unroll
        for (int v = 0; v < NV; v++) P0[v] = ldu_once(Pcb0 + v * ncb, o);
        E::PtoU(P0, dU, g);
      }
#endif
#else
      const bool u0 = false;
#endif
      if (!PLAIN && !prime && a.dE) {
        // calc_noRT_microphysics_dU (time_integrator.cpp:438-489): only the energy changes; k_cooling
        // left PtoU(p_new)[ERG] - PtoU(P)[ERG] of every domain cell (0 elsewhere)
        dU[uERG] += ldu(reinterpret_cast<const char *>(a.dE), pin_v(off_r));
      }

      // One task of a row: t = 0 the x face (c | c+1), t = 2 a y face, t = 3 the upper z face (c | c+sz).  The y task
      // has two modes: the upper face of the row (c | c+sy) and -- `lower`, first row of a group only -- its lower
      // face (c-sy | c), whose flux later rows get carried in registers.  Both modes run through ONE copy of the
      // code: the lower face of a group and the upper face of the group below are the same interface, solved by two
      // wavefronts, and from one copy both get the same bits in the fast build too (FMA contraction may differ from
      // copy to copy).
      // PION_ROWS2_COPIES: the task is a compile-time constant of each of three call sites (x, y, z), so the flux body
      // is inlined three times and the sweep-frame permutation, the edge-state arrays and what happens to the flux
      // need no selects and no copies at a loop join (first-order instance -6 %, second-order -1 %).
      // (read-only scalars are captured by value, the per-task temporaries live inside: a `cond ? a : b` on two
      // by-reference captures becomes a select of their addresses, which pins both to scratch memory)
      auto task = [&, c, off_r, offb_r, doff_n, doffb_n, r, row_ok, prime, k, noz, u0, jy_r, cy_m1, cy_0, cy_p1, cy_p2](auto tc, const int t_run, const bool lower) __attribute__((always_inline)) {
        constexpr int TC = decltype(tc)::value;
        const int t = (TC < 0) ? t_run : TC;
        double eL[NV], eR[NV], f[NV], pstar[NV];
        double s0y[NV];   // (CYL) this row's slope along R, sweep frame
        if constexpr (CYL) {
#pragma unroll
          for (int v = 0; v < NV; v++) s0y[v] = 0.0;
        }
        double bnm = 0.0, sim = 0.0, bnp = 0.0, sip = 0.0;   // B_n / psi of the lower and the upper neighbour cell
        long cl, st;
        int ax;
        unsigned hfl = 0, hfr = 0;   // HLLD -> HLL switch flags of the two cells of the interface
        const unsigned zt = opaque_zero();
        const char *const St = Sb + zt, *const Ht = Hb + zt;
        if (t == 0) {
          const unsigned off = pin_v(off_r), offb = pin_v(offb_r);   // (pin_v: offsets as this block sees them)
          ax = 0;
          st = 1;
          cl = c;
          if constexpr (MHD && SOLVER == FLUX_RS_HLLD) {
            if (PF && pf_valid) hfl = pfh;   // (both flags, packed)
            else {
              hfl = ldub(Ht, offb);
              hfr = ldub(Ht + 1, offb);
            }
          }
          if (oa2) {
            double qm[NV], qp[NV], sx[NV];
#pragma unroll
            for (int v = 0; v < NV; v++) {
              qm[v] = ldu(St + v * ncb - 8, off);
              qp[v] = ldu(St + v * ncb + 8, off);
            }
#if defined(PION_FAST_MATH) && PION_ROWS2_U0 == 2
            if (u0) {
              // the start-of-step state, into dU's registers (converted after the solve)
              const char *const Pcb0 = reinterpret_cast<const char *>(a.Pc) + zt;
#pragma unroll
              for (int v = 0; v < NV; v++) dU[v] = ldu_once(Pcb0 + v * ncb, off);
            }
#endif
            hslope3<NV>(qm, q0, qp, dx, thr, sx);
#pragma unroll
            for (int v = 0; v < NV; v++) {
              eL[v] = q0[v] + sx[v] * 0.5;
              const double em = q0[v] - sx[v] * 0.5;
              eR[v] = lane_next(em);
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
          else {
#pragma unroll
            for (int v = 0; v < NV; v++) {
              eL[v] = q0[v];
              eR[v] = lane_next(q0[v]);
            }
            if constexpr (MHD) {
              // first order: the x neighbours are only needed for B_n and psi, and those are the
              // neighbouring lanes' own cell values (the end lanes are halo lanes, their result is unused)
              bnm = lane_prev(q0[qBN]);
              bnp = eR[qBN];
              if constexpr (EQ == EQGLM) {
                sim = lane_prev(q0[qSI]);
                sip = eR[qSI];
              }
            }
          }
        }
        else if (t == 2) {
          // A y face between cell A (below) and cell B (above): upper mode A = this row, B = row j+1; lower mode
          // A = row j-1, B = this row.  F is the one of the two that has to be fetched, C the row above B.
          const unsigned off = pin_v(off_r), offb = pin_v(offb_r);   // (pin_v: offsets as this block sees them)
          ax = 1;
          st = sy;
          cl = lower ? c - sy : c;
          const long shc = lower ? -sy : 0, shb = lower ? -syb : 0;   // shift of cell A from the lane's cell (uniform)
          if constexpr (MHD && SOLVER == FLUX_RS_HLLD) {
            if (PF && pf_valid) hfl = pfh;
            else {
              hfl = ldub(Ht + shc, offb);
              hfr = ldub(Ht + shc + sy, offb);
            }
          }
          double own[NV], F[NV], A[NV], B[NV];
          if (PF && pf_valid) {
#pragma unroll
            for (int v = 0; v < NV; v++) F[v] = pf[v];
            bnm = pfb;
            sim = pfs;
          }
          else {
            load_rot2<NV, MHD>(St, ncb, 1, lower ? -syb : syb, off, F);
            if constexpr (MHD) {
              if (!lower) {
                bnm = ldu(St + (long)rotvar<MHD>(1, qBN) * ncb - syb, off);
                if constexpr (EQ == EQGLM) sim = ldu(St + (long)qSI * ncb - syb, off);
              }
            }
          }
          to_sweep<NV, MHD>(1, q0, own);
          if (lower) {
#pragma unroll
            for (int v = 0; v < NV; v++) {
              A[v] = F[v];
              B[v] = own[v];
            }
          }
          else {
#pragma unroll
            for (int v = 0; v < NV; v++) {
              A[v] = own[v];
              B[v] = F[v];
            }
          }
          if constexpr (CYL) {
            // (ysn carries the TRUE slope along R here; jA: all-cell y index of cell A)
            if (oa2) {
              const int jA = jy_r - (lower ? 1 : 0);
              const double RA = cyl_R(a.g, jA), RB = cyl_R(a.g, jA + 1);
              const double cA = lower ? cy_m1 : cy_0, cB = lower ? cy_0 : cy_p1, cC = lower ? cy_p1 : cy_p2;
              double C[NV], sB[NV];
              load_rot2<NV, MHD>(St, ncb, 1, shb + 2 * syb, off, C);
              if (lower) {
                double M[NV];
                load_rot2<NV, MHD>(St, ncb, 1, -2 * syb, off, M);
                cyl_slope3<NV>(M, A, B, cyl_Rcom(cyl_R(a.g, jA - 1), dx), cA, cB, true, ysn);
              }
              else {
                // this row's slope, for the geometric source after the solve
#pragma unroll
                for (int v = 0; v < NV; v++) s0y[v] = ysn[v];
              }
              cyl_slope3<NV>(A, B, C, cA, cB, cC, true, sB);
#pragma unroll
              for (int v = 0; v < NV; v++) {
                // VectorOps_Cyl::SetEdgeState (VectorOps.cpp:1052-1092): distance of the face from the centre of mass
                eL[v] = A[v] + ysn[v] * (RA + dx * 0.5 - cA);
                eR[v] = B[v] + sB[v] * (RB - dx * 0.5 - cB);
                ysn[v] = sB[v];
              }
            }
            else {
#pragma unroll
              for (int v = 0; v < NV; v++) {
                eL[v] = A[v];
                eR[v] = B[v];
              }
            }
          }
          else if (oa2) {
            double C[NV], sB[NV];
            load_rot2<NV, MHD>(St, ncb, 1, shb + 2 * syb, off, C);
            if (lower) {
              // the slope of row j-1 (an upper face finds its row's slope carried in ysn)
              double M[NV];
              load_rot2<NV, MHD>(St, ncb, 1, -2 * syb, off, M);
              hslope3<NV>(M, A, B, dx, thr, ysn);
            }
            hslope3<NV>(A, B, C, dx, thr, sB);
#pragma unroll
            for (int v = 0; v < NV; v++) {
              eL[v] = A[v] + ysn[v] * 0.5;
              eR[v] = B[v] - sB[v] * 0.5;
              ysn[v] = sB[v];   // the slope of B: of this row (lower mode), of the next row (upper mode)
            }
          }
          else {
#pragma unroll
            for (int v = 0; v < NV; v++) {
              eL[v] = A[v];
              eR[v] = B[v];
            }
          }
          if constexpr (MHD) {
            bnp = F[qBN];
            if constexpr (EQ == EQGLM) sip = F[qSI];
          }
        }
        else {
          // upper z face (c | c+sz)
          const unsigned off = pin_v(off_r), offb = pin_v(offb_r);   // (pin_v: offsets as this block sees them)
          ax = 2;
          st = sz;
          cl = c;
          if constexpr (MHD && SOLVER == FLUX_RS_HLLD) {
            if (PF && pf_valid) hfl = pfh;
            else {
              hfl = ldub(Ht, offb);
              hfr = ldub(Ht + sz, offb);
            }
          }
          double zq0[NV], qp1[NV];
          if (PF && pf_valid) {
#pragma unroll
            for (int v = 0; v < NV; v++) qp1[v] = pf[v];
            bnm = pfb;
            sim = pfs;
          }
          else {
            load_rot2<NV, MHD>(St, ncb, 2, szb, off, qp1);
            if constexpr (MHD) {
              bnm = ldu(St + (long)rotvar<MHD>(2, qBN) * ncb - szb, off);
              if constexpr (EQ == EQGLM) sim = ldu(St + (long)qSI * ncb - szb, off);
            }
          }
          to_sweep<NV, MHD>(2, q0, zq0);
          if (oa2) {
            double qp2[NV], sn[NV];
            load_rot2<NV, MHD>(St, ncb, 2, 2 * szb, off, qp2);
            hslope3<NV>(zq0, qp1, qp2, dx, thr, sn);
#pragma unroll
            for (int v = 0; v < NV; v++) eR[v] = qp1[v] - sn[v] * 0.5;
            if constexpr (ZSL) {
#pragma unroll
              for (int v = 0; v < NV; v++) {
                const double sc = ZS2(r, v);
                eL[v] = zq0[v] + sc * 0.5;
                ZS2(r, v) = sn[v];
              }
            }
            else {
              double qm1[NV], sc[NV];
              load_rot2<NV, MHD>(St, ncb, 2, -szb, off, qm1);
              hslope3<NV>(qm1, zq0, qp1, dx, thr, sc);
#pragma unroll
              for (int v = 0; v < NV; v++) eL[v] = zq0[v] + sc[v] * 0.5;
            }
          }
          else {
#pragma unroll
            for (int v = 0; v < NV; v++) {
              eL[v] = zq0[v];
              eR[v] = qp1[v];
            }
          }
          if constexpr (MHD) {
            bnp = qp1[qBN];
            if constexpr (EQ == EQGLM) sip = qp1[qSI];
          }
        }

        double hc_eta = 0.0;
        if (hcorr) hc_eta = select_hcorr_eta(a, ax, cl, st);
        const bool use_hll = (hfl | hfr) != 0;
        if (PF) {
          // requests for the task that follows this solve
          const unsigned zp = opaque_zero();
          const char *const Sp = Sb + zp, *const Hp = Hb + zp;
          // One load sequence for all cases, steered by uniform quantities (sweep axis of the rotation, row / plane
          // shift, which of the two cell offsets): written as loads inside the branches the optimiser merges them
          // into "load (phi of per-lane addresses)", i.e. 64-bit VALU address arithmetic per load.
          int pax;             // sweep axis whose frame the state is loaded in (0: natural order)
          long psh, bsh = 0;   // byte shift of the row to load; of the lower neighbour's B_n / psi
          long hs1, hs2;       // the two switch flags
          bool wb = true, nextcell = false;
          if (t == 3 || (noz && t == 2 && !lower)) {
            // next: the x task of the next row; in the priming plane, where every row runs the z task
            // only, the z task of the next row (after its last row: the x task of the first row of the
            // next plane)
            nextcell = true;
            if (prime && r + 1 < nrows) {
              pax = 2;
              psh = szb;
              bsh = -szb;
              hs1 = 0;
              hs2 = sz;
            }
            else {
              pax = 0;
              psh = 0;
              wb = false;
              hs1 = 0;
              hs2 = 1;
            }
          }
          else if (t == 0 && r == 0) {
            // next: the lower y face of the group's first row
            pax = 1;
            psh = -syb;
            wb = false;
            hs1 = -sy;
            hs2 = 0;
          }
          else if (t == 0 || lower) {
            // next: the upper y face of this row
            pax = 1;
            psh = syb;
            bsh = -syb;
            hs1 = 0;
            hs2 = sy;
          }
          else {   // an upper y face: next is the z task of this row
            pax = 2;
            psh = szb;
            bsh = -szb;
            hs1 = 0;
            hs2 = sz;
          }
          // (as arithmetic: a select between two captured values becomes a select of their addresses -> scratch)
          const unsigned off = pin_v(off_r + (nextcell ? doff_n : 0u)), offb = pin_v(offb_r + (nextcell ? doffb_n : 0u));
          load_rot2<NV, MHD>(Sp, ncb, pax, psh, off, pf);
          if constexpr (MHD) {
            if (wb) {
              pfb = ldu(Sp + (long)rotvar<MHD>(pax, qBN) * ncb + bsh, off);
              if constexpr (EQ == EQGLM) pfs = ldu(Sp + (long)qSI * ncb + bsh, off);
            }
          }
          if constexpr (MHD && SOLVER == FLUX_RS_HLLD) pfh = ldub(Hp + hs1, offb) | ldub(Hp + hs2, offb);
          pf_valid = true;
        }
        FX::intercell_flux(eL, eR, f, pstar, fc, hc_eta, use_hll, err);

        if (t == 0) {
          double Fm[NV];
#pragma unroll
          for (int v = 0; v < NV; v++) Fm[v] = lane_prev(f[v]);
#if defined(PION_FAST_MATH) && PION_ROWS2_U0 == 2
          if (u0) {
            double P0[NV];
#pragma unroll
            for (int v = 0; v < NV; v++) P0[v] = dU[v];
            E::PtoU(P0, dU, g);
          }
#endif
          apply_axis<EQ, NV>(dU, q0, bnm, sim, bnp, sip, Fm, f, dt, dx);
        }
        else if (t == 2) {
          if (!lower) {
            double d[NV], yq0[NV];
            to_sweep<NV, MHD>(1, q0, yq0);
            to_sweep<NV, MHD>(1, dU, d);
            if constexpr (CYL)
              apply_axis_cyl<EQ, NV>(d, yq0, bnm, sim, bnp, sip, Fy, f, s0y, dt, dx, cyl_R(a.g, jy_r - 1), cyl_R(a.g, jy_r), cy_0,
                                     oa2, a.fc.chyp);
            else apply_axis<EQ, NV>(d, yq0, bnm, sim, bnp, sip, Fy, f, dt, dx);
            from_sweep<NV, MHD>(1, d, dU);
          }
#pragma unroll
          for (int v = 0; v < NV; v++) Fy[v] = f[v];  // lower-face flux of this row (lower mode) / of the next row
        }
        else {
          if (!prime) {
            double d[NV], zq0[NV], Fzl[NV];
            to_sweep<NV, MHD>(2, q0, zq0);
            to_sweep<NV, MHD>(2, dU, d);
#pragma unroll
            for (int v = 0; v < NV; v++) Fzl[v] = ZS2(r, NZ - NV + v);
            apply_axis<EQ, NV>(d, zq0, bnm, sim, bnp, sip, Fzl, f, dt, dx);
            from_sweep<NV, MHD>(2, d, dU);
          }
#pragma unroll
          for (int v = 0; v < NV; v++) ZS2(r, NZ - NV + v) = f[v];
        }
      };
#if PION_ROWS2_COPIES
      if (!prime) {
        task(std::integral_constant<int, 0>{}, 0, false);
        // the y task: for the first row of the group its lower face first
#pragma unroll 1
        for (int m = (r == 0) ? 1 : 0; m >= 0; m--) task(std::integral_constant<int, 2>{}, 2, m != 0);
      }
      if (!noz) task(std::integral_constant<int, 3>{}, 3, false);
#else
#pragma unroll 1
      for (int t = prime ? 3 : 0; t < 4; t++) {
        if ((t == 1 && r > 0) || (t == 3 && noz)) continue;
        task(std::integral_constant<int, -1>{}, (t == 1) ? 2 : t, t == 1);
      }
#endif

      if (!prime && writer && row_ok) {
        const unsigned off = pin_v(off_r), offb = pin_v(offb_r);
        // (array bases made opaque here for the same reason as in the tasks: computed by the scalar unit in this
        // block instead of hoisted out of the loops, spilled to VGPR lanes and read back by VALU v_readlane)
        const unsigned zu = opaque_zero();
        const char *const Pcb = reinterpret_cast<const char *>(a.Pc) + zu;
        char *const Ob = reinterpret_cast<char *>(a.out) + zu;
        // (a grid without internal-boundary cells: every on-grid cell is isgd | isdomain | timestep | isleaf)
        const unsigned fl = (PLAIN && a.plain_cells) ? 29u : ldub(reinterpret_cast<const char *>(a.flags) + zu, offb);
        double P0[NV], Pf[NV];
        if (u0) {
          // dU already holds PtoU(P0) + dU: the tail of cell_update (CellAdvanceTime) from there
          E::UtoP(dU, Pf, a.fc.min_temp, g, MPd{}, err);
          if constexpr (EQ == EQGLM) Pf[qSI] *= a.glm_damp;
        }
        else {
        if (same_pc) {
#pragma unroll
          for (int v = 0; v < NV; v++) P0[v] = q0[v];
        }
        else {
#pragma unroll
          for (int v = 0; v < NV; v++) P0[v] = ldu_once(Pcb + v * ncb, off);
        }
        if (!(fl & 4) || !(fl & 16)) {
#pragma unroll
          for (int v = 0; v < NV; v++) Pf[v] = P0[v];
        }
        else cell_update<EQ, NTR>(a, P0, dU, err, Pf, PLAIN);
        }
#pragma unroll
        for (int v = 0; v < NV; v++) stu(Ob + v * ncb, off, Pf[v]);
        if (a.xwrap) {
          // periodic x faces: the ghost images of the first / last nbc cells of the row are these same values
          // (periodic_boundaries.cpp:42-50); written here, where the row is in registers, instead of by the
          // boundary kernel, whose x slab is one uncoalesced 8-byte access per row and variable
          const long xs = (long)a.g.ng[0] * 8;
          if (ix < a.g.nbc[0]) {
#pragma unroll
            for (int v = 0; v < NV; v++) stu(Ob + v * ncb + xs, off, Pf[v]);
          }
          else if (ix >= a.g.ng[0] - a.g.nbc[0]) {
#pragma unroll
            for (int v = 0; v < NV; v++) stu(Ob + v * ncb - xs, off, Pf[v]);
          }
        }
        if (a.dtres) {
          // calc_dynamics_dt / calc_microphysics_dt (calc_timestep.cpp:271-507) of the state just
          // written: after a full step it is the state the next step's dt is computed from
          if ((fl & 8) && !(fl & 2)) {
            const double t = cell_dt<EQ>(Pf, a.g.ndim, g, dx, a.cfl);
            if (!(t > 0.0)) err |= ERR_BAD_DT;
            tdyn = (t < tdyn) ? t : tdyn;
          }
          if (a.dt_mp && !(fl & 2) && (fl & 16)) {
            const double t = Cooling::timescale(a.cool, Pf[qRO], Pf[qPG], g);
            tmp = (t < tmp) ? t : tmp;
          }
        }
      }
    }
  }
#undef ZS2
  if (a.dtres) {
    tdyn = wave_min64(tdyn);
    tmp = wave_min64(tmp);
    if (lane == 0) {
      atomicMin(&a.dtres[0], (unsigned long long)__double_as_longlong(tdyn));
      atomicMin(&a.dtres[1], (unsigned long long)__double_as_longlong(tmp));
    }
  }
  if (err) atomicOr(a.errword, err);
}

// rows per wavefront the LDS budget allows
template <int NV, bool ZSL>
__host__ inline int rows2_rmax()
{
  constexpr int NZ = ZSL ? 2 * NV : NV;
  int r = (int)(PION_ROWS2_LDS_BYTES / (sizeof(double) * 4 * NZ * 64));
  return r > 8 ? 8 : r;
}

// 2-D launches: rows per wavefront for THIS instance.  A 2-D launch is one to a few "rounds" of wavefronts (one
// wavefront marches its R rows from start to end), so what matters is how well the wavefronts fill the slots of the
// rounds they need: R is the value in [8, 64] with the best (filled share of the slots) / (2 + 1/R Riemann solves per
// cell) -- for launches of at most three rounds at the caller's R; the slots follow from the occupancy of the instance
// (its registers).  Measured, Euler Roe-CV 4096 x 1260
// (66 x-tiles, 3 wavefronts per SIMD): R = 16 (1.7 rounds) 12 790, 28 (0.97 of one round) 14 240-14 300, 32 13 150
// Mcell-updates/s.  The result does not depend on R (tests/test_gpu_xtile.py).
template <void (*KERNEL)(const StageArgs)>
static int rows2_pick_rows_2d(const StageArgs &a)
{
  static int wg_per_cu = 0;   // (per instance)
  if (wg_per_cu == 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, KERNEL, 256, 0) != hipSuccess || n < 1) n = 2;
    wg_per_cu = n;
  }
  const long slots = 4L * wg_per_cu * (a.ncu > 0 ? a.ncu : 256);
  int best = a.rows;
  double best_score = -1.0;
  StageArgs t = a;
  // (a launch of many rounds is filled well enough at the caller's rows, and long columns cost it L2 locality:
  // 4096 x 6144 GLM-MHD HLLD, 12.4 rounds at R = 16: 8340 Mcell-updates/s, R = 50 -- one round fewer -- 7350)
  if ((rows_tiling(a).per_chunk + slots - 1) / slots > 3) return a.rows;
  for (int R = 8; R <= 64; R++) {
    t.rows = R;
    const long waves = rows_tiling(t).per_chunk;
    const long rounds = (waves + slots - 1) / slots;
    const double score = ((double)waves / (double)(rounds * slots)) / (2.0 + 1.0 / R);
    if (score > best_score * 1.0000001) {
      best_score = score;
      best = R;
    }
  }
  return best;
}
// one launch of an instance: rows per wavefront (2-D: picked here when the caller leaves the choice), grid, LDS
template <void (*KERNEL)(const StageArgs)>
static int rows2_launch(StageArgs a, const int rmax, const size_t lds_bytes_per_row, hipStream_t s)
{
  const bool noz = (a.g.ndim == 2);
  if (noz && a.rows_auto) a.rows = rows2_pick_rows_2d<KERNEL>(a);
  if (a.rows > rmax) a.rows = rmax;
  if (a.rows < 1) a.rows = 1;
  const int nzc = ((a.nzb > 0) ? a.nzb : (a.kz1 - a.kz0 + a.zchunk - 1) / a.zchunk) + (a.kz3 - a.kz2 + a.zchunk - 1) / a.zchunk;
  const int nb4 = (rows_tiling(a).per_chunk + 3) / 4, nb8 = (nb4 + 7) / 8;
  const long nblocks = 8L * nb8 * nzc;   // (see the kernel: an eighth of the x-y tiles per XCD, through all chunks)
  const size_t shmem = noz ? 0 : lds_bytes_per_row * a.rows;
  hipLaunchKernelGGL(KERNEL, dim3((unsigned)nblocks), dim3(256), shmem, s, a);
  return (int)hipGetLastError();
}

template <int EQ, int NTR, int SOLVER, bool ZSL>
static int stage_rows2_go_z(const StageArgs &a, hipStream_t s)
{
  constexpr int NV = Eqn<EQ, NTR>::NV;
  constexpr int NZ = ZSL ? 2 * NV : NV;
  const bool noz = (a.g.ndim == 2);
  const int rmax = noz ? 64 : rows2_rmax<NV, ZSL>();   // (2-D: nothing is carried in LDS)
  constexpr size_t lds_row = sizeof(double) * 4 * NZ * 64, lds_row1 = sizeof(double) * 4 * NV * 64;
  const int rmax1 = noz ? 64 : rows2_rmax<NV, false>();
  // compile-time spatial order and "no H-correction / microphysics" for the production instances
  // (MHD HLLD, Euler Roe-CV, Euler FVS), run-time for the others
  constexpr bool specialise = (SOLVER == FLUX_RS_HLLD && EQ != EQEUL)
                              || ((SOLVER == FLUX_RSroe || SOLVER == FLUX_FVS) && EQ == EQEUL);
  if constexpr (specialise) {
    const bool plain = (a.cooling == 0 && !a.fc.mp.present && a.fc.artvisc != AV_HCORRECTION
                        && a.fc.artvisc != AV_HCORR_FKJ98);
    if (plain) {
      if (a.space_ooa == 2) return rows2_launch<k_stage_rows2<EQ, NTR, SOLVER, 2, true, ZSL>>(a, rmax, lds_row, s);
      return rows2_launch<k_stage_rows2<EQ, NTR, SOLVER, 1, true, false>>(a, rmax1, lds_row1, s);
    }
    if (a.space_ooa == 2) return rows2_launch<k_stage_rows2<EQ, NTR, SOLVER, 2, false, ZSL>>(a, rmax, lds_row, s);
    return rows2_launch<k_stage_rows2<EQ, NTR, SOLVER, 1, false, false>>(a, rmax1, lds_row1, s);
  }
  else
    return rows2_launch<k_stage_rows2<EQ, NTR, SOLVER, 0, false, ZSL>>(a, rmax, lds_row, s);
}

// cylindrical (z,R) 2-D grids: the CYL instances (the same specialisation rule, no LDS)
template <int EQ, int NTR, int SOLVER>
static int stage_rows2_go_cyl(const StageArgs &a, hipStream_t s)
{
  constexpr bool specialise = (SOLVER == FLUX_RS_HLLD && EQ != EQEUL)
                              || ((SOLVER == FLUX_RSroe || SOLVER == FLUX_FVS) && EQ == EQEUL);
  if constexpr (specialise) {
    const bool plain = (a.cooling == 0 && !a.fc.mp.present && a.fc.artvisc != AV_HCORRECTION
                        && a.fc.artvisc != AV_HCORR_FKJ98);
    if (plain && a.space_ooa == 2) return rows2_launch<k_stage_rows2<EQ, NTR, SOLVER, 2, true, false, true>>(a, 64, 0, s);
    if (plain && a.space_ooa == 1) return rows2_launch<k_stage_rows2<EQ, NTR, SOLVER, 1, true, false, true>>(a, 64, 0, s);
  }
  return rows2_launch<k_stage_rows2<EQ, NTR, SOLVER, 0, false, false, true>>(a, 64, 0, s);
}

template <int EQ, int NTR, int SOLVER>
static int stage_rows2_go(const StageArgs &a, hipStream_t s)
{
  if (a.g.ndim == 2 && a.g.cyl == 1) return stage_rows2_go_cyl<EQ, NTR, SOLVER>(a, s);
  // a.zslope_lds: carry the z slope in LDS (fewer rows per wavefront) instead of rebuilding it
  if (a.zslope_lds && a.space_ooa == 2) return stage_rows2_go_z<EQ, NTR, SOLVER, true>(a, s);
  return stage_rows2_go_z<EQ, NTR, SOLVER, false>(a, s);
}

#endif
