// rows_tiling.h -- x/y tiling of one plane chunk of the 3-D stage kernel (k_stage_rows2, stage_rows2.h),
// shared by the kernel and its launcher.
//
// x: full tiles of PION_MARCH_XT = 62 cells, one wavefront each (lanes 0 and 63 are halo lanes that only
// supply their neighbour's interface data).  The remaining `rem` cells of a row (16 of 512) would leave most
// of a wavefront idle, so a "remainder" wavefront packs the remainders of `spw` consecutive row groups side by
// side, each in a segment of rem + 2 lanes with its own two halo lanes (the x shuffles only ever cross a segment
// boundary into a halo lane).  y: groups of a.rows consecutive rows per wavefront.
#ifndef PION_ROWS_TILING_H
#define PION_ROWS_TILING_H

// x/y tiling of one plane chunk, shared by the kernel and its launcher
struct RowsTiling {
  int ntx_full, rem, spw, nyg, nfull, nrem, per_chunk;
};
__host__ __device__ inline RowsTiling rows_tiling(const StageArgs &a)
{
  RowsTiling t;
  t.nyg = (a.g.ng[1] + a.rows - 1) / a.rows;
  t.ntx_full = a.g.ng[0] / PION_MARCH_XT;
  t.rem = a.g.ng[0] - t.ntx_full * PION_MARCH_XT;
  t.spw = (t.rem > 0) ? 64 / (t.rem + 2) : 0;   // row groups per remainder wavefront
  t.nfull = t.ntx_full * t.nyg;
  t.nrem = (t.rem > 0) ? (t.nyg + t.spw - 1) / t.spw : 0;
  t.per_chunk = t.nfull + t.nrem;
  return t;
}

#endif
