// kernels.h -- launch interface between the C-ABI layer (pion_gpu.hip) and the
// floating-point kernels (kernels_fp.hip).  kernels_fp.hip is compiled twice,
// once per floating-point mode, into namespaces pion::fp_strict (-ffp-contract=off:
// bit-parity with the reference's x86-64 -O3 build) and pion::fp_fast (FMA
// contraction allowed); pion_gpu_config::strict_fp selects at run time.
#ifndef PION_KERNELS_H
#define PION_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev_riemann.h"

namespace pion {

#define PION_MAX_NTR 2
// LDS a workgroup of k_stage_rows2 may use so that two fit a CU (160 KiB)
#define PION_ROWS2_LDS_BYTES (80 * 1024)
#define PION_COOL_NT_MAX 256

struct GridDesc {
  int ndim;
  int ng[3], nbc[3], nga[3];
  long ncell;   // cells incl. ghosts
  long sy, sz;  // strides of y and z in cells
  double dx;
  double xmin[3];
  int cyl;      // 1: cylindrical (z,R) axisymmetry, axis 1 = R (2-D only); 2: spherical symmetry, axis 0 = R (1-D)
  const double *sph_vol;  // spherical: (rp^3 - rn^3)/3 per all-cell x index, evaluated on the host (libm pow)
};

struct CoolDev {
  int NT;
  const double *T;       // [NT]
  const double *tab;     // [5][NT]  rrhp, C_rrh, C_ffhe, C_fbdn, C_cie
  const double *slope;   // [5][NT]
  double inv_Mu2, inv_Mu2_elec_H, Mu_tot_over_kB;
  double MinT_allowed, MaxT_allowed;
  // log-spaced temperature grid (what gen_mpoc_lookup_tables builds): first guess of the table interval,
  // (log2 T - lg0) * inv_dlg, corrected against the table itself; inv_dlg = 0: not log-spaced, bisect
  float lg0, inv_dlg;
};

// Uneven plane chunks of a strip of np planes: chunk number cz covers [*k0, *k1) (relative to the strip); returns the
// number of chunks.  A rule instead of a table in the kernel arguments: indexing an argument array with a run-time
// index makes the compiler copy the whole argument struct to scratch memory (measured: the stage kernel 2x slower).
__host__ __device__ inline int zchunk_bounds(const int np, const int cmax, const int cz, int *k0, int *k1)
{
  int n = 0, pos = 0;
  *k0 = *k1 = np;
  while (pos < np) {
    const int rem = np - pos;
    const int cmin = (np <= 128) ? 2 : 4;   // (a slab of a few dozen planes: its tail is a larger share of the launch)
    int c = (rem > 2 * cmax) ? cmax : ((rem / 2 > cmin) ? rem / 2 : cmin);
    if (c > cmax) c = cmax;
    if (c > rem || rem - c < cmin) c = rem;
    if (n == cz) {
      *k0 = pos;
      *k1 = pos + c;
    }
    pos += c;
    n++;
  }
  return n;
}
struct StageArgs {
  GridDesc g;
  const double *S;    // stencil state ("Ph")      [nvar][ncell]
  const double *Pc;   // start-of-step state ("P") [nvar][ncell]
  double *out;        // destination                [nvar][ncell]
  const uint8_t *flags;
  const uint8_t *hllflag;   // HLLD->HLL switch per cell (may be null)
  const double *eta;        // H-correction [ndim][ncell] (may be null)
  int *errword;
  FluxCtx fc;
  int eqntype, ntracer, solver;
  int space_ooa;
  int cooling;        // EP.cooling
  double dt;          // stage dt (= FV_dt)
  double glm_damp;    // exp(-FV_dt*chyp*cr), evaluated on the host
  double max_temp;    // EP.MaxTemperature
  int use_march;      // != 0: k_stage_rows2 (3-D, nbc >= 2: production), 0: k_stage (cell per thread; 1-D / 2-D, cross-check)
  int xwrap;          // k_stage_rows2: x faces periodic -> also write the x ghost images of the rows it updates
  int zslope_lds;     // k_stage_rows2: carry the z slope in LDS (else rebuild it from plane k-1)
  double *dE;         // k_stage_rows2: cooling source PtoU(p_new)[ERG]-PtoU(P)[ERG] per cell from k_cooling_dE (or null)
  int zchunk;         // planes per wavefront in the marching kernels
  // k_stage_rows2, first strip [kz0,kz1): nzb > 0 = uneven chunks (zchunk_bounds below: chunks of `zcmax` planes,
  // then halving down to 4 -- long chunks first, short ones last: the launch's last wavefronts are short, so its
  // tail is), nzb of them
  int nzb, zcmax;
  int rows;           // y-rows per wavefront in k_stage_rows2
  int rows_auto;      // 2-D: the launcher may pick the rows per wavefront for the instance it launches (its occupancy)
  int ncu;            // compute units of the device (for that choice)
  int kz0, kz1;       // on-grid z planes [kz0,kz1) this launch updates (k_stage_rows2; k_stage: whole grid)
  int kz2, kz3;       // and a second strip [kz2,kz3) (empty when kz3 <= kz2): the two z-boundary strips
  unsigned long long *dtres;  // k_stage_rows2, full step: min t_dyn / t_mp bits of the new state (or null)
  double cfl;
  int dt_mp;          // also reduce the cooling time (EP.MP_timestep_limit)
  int plain_cells;    // every on-grid cell is an ordinary domain cell (no stellar-wind cells): flags need not be read
  CoolDev cool;
};

struct PrepassArgs {
  GridDesc g;
  const double *S;
  uint8_t *hllflag;
  double *divv, *gradp;   // optional debug outputs (null in production)
  double *eta;            // [ndim][ncell]
  int eqntype, nvar, space_ooa;
  double gamma;
  long c0, c1;            // cell range [c0,c1) (whole planes) this launch covers
  long c2, c3;            // k_prepass_hlld: a second range [c2,c3) in the same launch (empty when c3 <= c2)
};

struct DtArgs {
  GridDesc g;
  const double *P;
  const double *Ph;
  const uint8_t *flags;
  unsigned long long *result;  // [0]=t_dyn bits, [1]=t_mp bits (min over positive doubles)
  int *errword;
  int eqntype, nvar;
  double gamma, cfl;
  int do_mp;
  CoolDev cool;
};

struct FluxTestArgs {
  int n, axis, eqntype, ntracer, solver;
  const double *Pl, *Pr, *aux;
  double *F, *Pstar;
  int *errword;
  FluxCtx fc;
};

struct CoolTestArgs {
  int n, nvar;
  double dt, gamma;
  const double *Pin;
  double *Pout;
  const double *rho, *T;
  double *edot;
  int *errword;
  CoolDev cool;
};

#define PION_DECLARE_FP(NS)                                        \
  namespace NS {                                                   \
  int launch_stage(const StageArgs &a, hipStream_t s);             \
  int launch_cooling_dE(const StageArgs &a, hipStream_t s);        \
  int stage_rows2_rows(int eq, int ntr, int zslope_lds, int want); \
  int launch_prepass(const PrepassArgs &a, hipStream_t s);         \
  int launch_dt(const DtArgs &a, hipStream_t s);                   \
  int launch_dt_mp(const DtArgs &a, hipStream_t s);                \
  int launch_flux_test(const FluxTestArgs &a, hipStream_t s);      \
  int launch_cool_update(const CoolTestArgs &a, hipStream_t s);    \
  int launch_cool_edot(const CoolTestArgs &a, hipStream_t s);      \
  int launch_cool_timescale(const CoolTestArgs &a, hipStream_t s); \
  const char *stage_kernel_name(int eq, int ntr, int solver);      \
  }

PION_DECLARE_FP(fp_strict)
PION_DECLARE_FP(fp_fast)

}  // namespace pion
#endif
