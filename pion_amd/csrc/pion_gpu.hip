// pion_gpu.hip -- implementation of the C-ABI declared in include/pion_gpu.h.
//
// Host side of the boundary: owns device memory behind an opaque handle, launches
// the floating-point kernels of kernels_fp.hip (strict or fast namespace), and
// holds the data-movement kernels that have no arithmetic: ghost-cell fills
// (boundaries/*.cpp of the reference), stellar-wind cell reset, halo pack/unpack.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pion_gpu.h"
#include "kernels.h"

using namespace pion;

namespace {

struct BCArgs {
  GridDesc g;
  double *T;        // array whose ghosts are filled (sources are read from the same array)
  int nvar, dir, type, eqntype, ntracer;
  double refval[PION_MAX_NVAR];
  double dmr_a0, dmr_t3;  // 10*simtime/sin(pi/3), tan(pi/3)  (host libm, as the reference)
};

// All periodic faces of a grid in ONE launch (the bench configuration).  The X -> Y -> Z sequence of
// periodic copies (periodic_boundaries.cpp:42-50, corner cells through already filled ghosts) ends with
// every ghost cell holding the on-grid cell at its coordinates wrapped axis by axis, so the wrapped
// cell can be read directly: the same values, one launch instead of six.  z faces of kind SLAB
// (neighbour rank) are left alone: then only ghosts on on-grid z planes are filled.
__global__ __launch_bounds__(256) void k_bc_periodic_all(double *T, const GridDesc g, const int nvar, const int zwrap,
                                                         const int skipx)
{
  // ghost cells as three disjoint slabs: A = z ghosts (all x,y), B = y ghosts on on-grid z (all x),
  // C = x ghosts on on-grid y and z
  const long nA = zwrap ? (long)2 * g.nbc[2] * g.nga[0] * g.nga[1] : 0;
  const long nB = (long)g.ng[2] * 2 * g.nbc[1] * g.nga[0];
  // (skipx: the stage kernel has already written the x ghosts of the on-grid rows, slab C)
  const long nC = skipx ? 0 : (long)g.ng[2] * g.ng[1] * 2 * g.nbc[0];
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nA + nB + nC) return;
  int i0, i1, i2;  // all-cell coordinates (ghosts included)
  if (t < nA) {
    i0 = (int)(t % g.nga[0]);
    i1 = (int)((t / g.nga[0]) % g.nga[1]);
    const int kz = (int)(t / ((long)g.nga[0] * g.nga[1]));
    i2 = (kz < g.nbc[2]) ? kz : g.ng[2] + kz;
  }
  else if (t < nA + nB) {
    t -= nA;
    i0 = (int)(t % g.nga[0]);
    const int ky = (int)((t / g.nga[0]) % (2 * g.nbc[1]));
    i1 = (ky < g.nbc[1]) ? ky : g.ng[1] + ky;
    i2 = (int)(t / ((long)g.nga[0] * 2 * g.nbc[1])) + g.nbc[2];
  }
  else {
    t -= nA + nB;
    const int kx = (int)(t % (2 * g.nbc[0]));
    i0 = (kx < g.nbc[0]) ? kx : g.ng[0] + kx;
    i1 = (int)((t / (2 * g.nbc[0])) % g.ng[1]) + g.nbc[1];
    i2 = (int)(t / ((long)2 * g.nbc[0] * g.ng[1])) + g.nbc[2];
  }
  // wrap each coordinate back onto the grid
  int s0 = i0, s1 = i1, s2 = i2;
  if (s0 < g.nbc[0]) s0 += g.ng[0];
  else if (s0 >= g.nbc[0] + g.ng[0]) s0 -= g.ng[0];
  if (s1 < g.nbc[1]) s1 += g.ng[1];
  else if (s1 >= g.nbc[1] + g.ng[1]) s1 -= g.ng[1];
  if (zwrap) {
    if (s2 < g.nbc[2]) s2 += g.ng[2];
    else if (s2 >= g.nbc[2] + g.ng[2]) s2 -= g.ng[2];
  }
  const long c = (long)i0 + g.sy * i1 + g.sz * i2, sc = (long)s0 + g.sy * s1 + g.sz * s2;
  for (int v = 0; v < nvar; v++) T[v * g.ncell + c] = T[v * g.ncell + sc];
}

// Every external face of a grid in ONE launch, any mix of boundary types (what k_bc_periodic_all does for the
// all-periodic case).  The reference updates the faces one after the other in list order XN, XP, YN, YP, ZN, ZP
// (assign_update_bcs.cpp:185-252); a ghost cell belongs to the list of the HIGHEST axis along which it is a
// ghost (X lists hold on-grid (y,z) rows, Y lists the full x extent, Z lists the full x-y extent,
// uniform_grid.cpp:1009-1216), and a corner ghost takes its value from a ghost cell that a lower axis's update has
// just filled.  That chain of copies always ends on an on-grid cell (or on a constant state), which no boundary
// update writes: so each thread walks the chain of ITS ghost cell down the axes, reads the terminal cell, and
// applies the per-face operations (sign flips, one-way clamp, psi rule) on the way back up, lowest axis first --
// the same values as the six launches, without their ordering.  psi of GLM-MHD follows its own chain: outflow
// and one-way faces take -psi of the MIRROR cell (outflow_boundaries.cpp:140-152), everything else of the copy
// source.  Then the internal DMR2 boundary, which the reference applies last (double_Mach_ref_boundaries.cpp:98-147).
// z faces of kind SLAB (neighbour rank) are left alone.
struct BCAllArgs {
  GridDesc g;
  double *T;
  int nvar, eqntype, ntracer, ndim;
  int type[6];
  double refval[6][PION_MAX_NVAR];
  double dmr_a0, dmr_t3;
  int dmr2_cols;              // > 0: internal DMR2 boundary over the first dmr2_cols on-grid columns
  double dmr2_val[PION_MAX_NVAR];
};

__global__ __launch_bounds__(256) void k_bc_all(const BCAllArgs a)
{
  const GridDesc &g = a.g;
  const int nd = a.ndim;
  // ghost cells as three disjoint slabs: A = z ghosts (all x,y), B = y ghosts on on-grid z (all x),
  // C = x ghosts on on-grid y and z
  const long nA = (nd == 3) ? (long)2 * g.nbc[2] * g.nga[0] * g.nga[1] : 0;
  const long nB = (nd >= 2) ? (long)g.ng[2] * 2 * g.nbc[1] * g.nga[0] : 0;
  const long nC = (long)g.ng[2] * g.ng[1] * 2 * g.nbc[0];
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nA + nB + nC) return;
  int i[3];   // all-cell coordinates (ghosts included)
  if (t < nA) {
    i[0] = (int)(t % g.nga[0]);
    i[1] = (int)((t / g.nga[0]) % g.nga[1]);
    const int kz = (int)(t / ((long)g.nga[0] * g.nga[1]));
    i[2] = (kz < g.nbc[2]) ? kz : g.ng[2] + kz;
  }
  else if (t < nA + nB) {
    t -= nA;
    i[0] = (int)(t % g.nga[0]);
    const int ky = (int)((t / g.nga[0]) % (2 * g.nbc[1]));
    i[1] = (ky < g.nbc[1]) ? ky : g.ng[1] + ky;
    i[2] = (int)(t / ((long)g.nga[0] * 2 * g.nbc[1])) + g.nbc[2];
  }
  else {
    t -= nA + nB;
    const int kx = (int)(t % (2 * g.nbc[0]));
    i[0] = (kx < g.nbc[0]) ? kx : g.ng[0] + kx;
    i[1] = (int)((t / (2 * g.nbc[0])) % g.ng[1]) + g.nbc[1];
    i[2] = (int)(t / ((long)2 * g.nbc[0] * g.ng[1])) + g.nbc[2];
  }
  const long nc = g.ncell;
  const long c = (long)i[0] + g.sy * i[1] + g.sz * i[2];
  const bool mhd = (a.eqntype == EQMHD || a.eqntype == EQGLM);
  const bool glm = (a.eqntype == EQGLM);

  // ---- down the axes: the chain of source cells (s: all variables but psi; p: psi)
  int s[3] = {i[0], i[1], i[2]}, p[3] = {i[0], i[1], i[2]};
  int op_type[3] = {0, 0, 0}, op_pos[3] = {0, 0, 0};
  int const_ax = -1;   // axis whose face gives a constant / analytic state: the chain ends there
  bool owned = false;
  for (int ax = nd - 1; ax >= 0; ax--) {
    const int lo = g.nbc[ax], hi = g.nbc[ax] + g.ng[ax];
    if (s[ax] >= lo && s[ax] < hi) continue;   // on-grid along this axis
    const bool pos = (s[ax] >= hi);
    const int type = a.type[2 * ax + (pos ? 1 : 0)];
    // the first ghost axis met is the list this cell belongs to: a SLAB (neighbour rank) or unset face there
    // means the cell is not ours to fill
    if (!owned && (type == 0 || type == PION_BC_SLAB)) return;
    owned = true;
    op_type[ax] = type;
    op_pos[ax] = pos ? 1 : 0;
    const int depth = pos ? s[ax] - hi + 1 : lo - s[ax];   // distance from the grid = -isedge
    if (type == PION_BC_PERIODIC) {
      s[ax] += pos ? -g.ng[ax] : g.ng[ax];
      p[ax] = s[ax];
    }
    else if (type == PION_BC_INFLOW || type == PION_BC_FIXED || type == PION_BC_DMACH) {
      const_ax = ax;
      break;
    }
    else if (type == 0 || type == PION_BC_SLAB) {
      // (an unset face below the owning axis: the source is that ghost cell as it stands)
      op_type[ax] = 0;
      break;
    }
    else {
      // outflow, one-way, reflecting, axisymmetric, jet-reflect: every ghost layer copies the FIRST on-grid cell
      // of the row (outflow_boundaries.cpp:50-59); psi of an outflow / one-way face: the mirror cell
      s[ax] = pos ? hi - 1 : lo;
      if (glm && (type == PION_BC_OUTFLOW || type == PION_BC_ONEWAY_OUT)) p[ax] = pos ? hi - depth : lo + depth - 1;
      else p[ax] = s[ax];
    }
  }

  // ---- the terminal state
  double val[PION_MAX_NVAR];
  int from = 0;   // first axis whose operation is applied on the way up
  if (const_ax >= 0) {
    const int d = 2 * const_ax + op_pos[const_ax];
    if (op_type[const_ax] == PION_BC_DMACH) {
      // double_Mach_ref_boundaries.cpp:168-204, position of the ghost cell of the Y list (x may be a ghost)
      const int ix = s[0] - g.nbc[0], iy = s[1] - g.nbc[1];
      const double x = g.xmin[0] + (2 * ix + 1) * (0.5 * g.dx);
      const double y = g.xmin[1] + (2 * iy + 1) * (0.5 * g.dx);
      const double bpos = a.dmr_a0 + 1.0 / 6.0 + y / a.dmr_t3;
      if (x <= bpos) {
        val[0] = 8.0;
        val[1] = 116.5;
        val[2] = 7.14470958;
        val[3] = -4.125;
        val[4] = 0.0;
        for (int v = 5; v < a.nvar; v++) val[v] = 0.0;
        for (int v = a.nvar - a.ntracer; v < a.nvar; v++) val[v] = 1.0;
      }
      else {
        for (int v = 0; v < a.nvar; v++) val[v] = a.refval[d][v];
      }
    }
    else {
      for (int v = 0; v < a.nvar; v++) val[v] = a.refval[d][v];
    }
    from = const_ax + 1;
  }
  else {
    const long sc = (long)s[0] + g.sy * s[1] + g.sz * s[2];
    for (int v = 0; v < a.nvar; v++) val[v] = a.T[v * nc + sc];
    if (glm) {
      const long pc = (long)p[0] + g.sy * p[1] + g.sz * p[2];
      if (pc != sc) val[8] = a.T[8 * nc + pc];
    }
  }

  // ---- back up: the operations of the faces, lowest axis first (the order the reference applies them in)
  for (int ax = from; ax < nd; ax++) {
    const int type = op_type[ax];
    if (type == PION_BC_REFLECTING) {
      // reflecting_boundaries.cpp:34-73,131-153: normal velocity (and normal B) flip sign
      val[2 + ax] = val[2 + ax] * -1.0;
      if (mhd) val[5 + ax] = val[5 + ax] * -1.0;
    }
    else if (type == PION_BC_AXISYMMETRIC) {
      // axisymmetric_boundaries.cpp:34-52,98-137 (R = 0 axis): the radial and the theta components
      val[3] = val[3] * -1.0;
      val[4] = val[4] * -1.0;
      if (mhd) {
        val[6] = val[6] * -1.0;
        val[7] = val[7] * -1.0;
      }
    }
    else if (type == PION_BC_JETREFLECT) {
      // jetreflect_boundaries.cpp:32-62: v_n and the two tangential field components
      val[2 + ax] = val[2 + ax] * -1.0;
      if (mhd)
        for (int v = 5; v <= 7; v++)
          if (v != 5 + ax) val[v] = val[v] * -1.0;
    }
    else if (type == PION_BC_OUTFLOW || type == PION_BC_ONEWAY_OUT) {
      if (type == PION_BC_ONEWAY_OUT) {
        // oneway_out_boundaries.cpp:75-138
        const double sg = op_pos[ax] ? 1.0 : -1.0;
        const double x = val[2 + ax] * sg;
        val[2 + ax] = sg * ((0.0 < x) ? x : 0.0);
      }
      if (glm) val[8] = -val[8];   // GLM_NEGATIVE_BOUNDARY (boundaries.h:21)
    }
  }
  // internal DMR2 boundary: y < 0 ghost cells above the first on-grid columns (x <= 1/6) hold a fixed state
  if (a.dmr2_cols > 0 && i[1] < g.nbc[1] && i[0] >= g.nbc[0] && i[0] < g.nbc[0] + a.dmr2_cols) {
    for (int v = 0; v < a.nvar; v++) val[v] = a.dmr2_val[v];
  }
  for (int v = 0; v < a.nvar; v++) a.T[v * nc + c] = val[v];
}

// One thread per ghost cell of one face.  List membership follows UniformGrid::SetupBCs
// (grid/uniform_grid.cpp:1009-1216): X faces hold on-grid (y,z) rows only, Y faces the full x
// extent, Z faces the full x-y extent, which together with the X->Y->Z launch order fills the
// corner ghosts as the reference does.
__global__ __launch_bounds__(256) void k_bc_face(const BCArgs a)
{
  const int ax = a.dir / 2;
  const bool pos = a.dir & 1;
  int lo[3], n[3];
  for (int d = 0; d < 3; d++) {
    if (d == ax) {
      lo[d] = 0;
      n[d] = a.g.nbc[ax];
    }
    else if (d < ax || d >= a.g.ndim) {
      lo[d] = -a.g.nbc[d];
      n[d] = a.g.nga[d];
    }
    else {
      lo[d] = 0;
      n[d] = a.g.ng[d];
    }
  }
  const long total = (long)n[0] * n[1] * n[2];
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  int i[3];
  i[0] = (int)(t % n[0]) + lo[0];
  i[1] = (int)((t / n[0]) % n[1]) + lo[1];
  i[2] = (int)(t / ((long)n[0] * n[1])) + lo[2];
  // along the face axis the local index 0..nbc-1 becomes the ghost coordinate
  const int k = i[ax];
  const int depth = pos ? k + 1 : a.g.nbc[ax] - k;  // distance from the grid = -isedge
  i[ax] = pos ? a.g.ng[ax] + k : -a.g.nbc[ax] + k;
  const long nc = a.g.ncell;
  const long st = (ax == 0) ? 1 : ((ax == 1) ? a.g.sy : a.g.sz);
  const long c = (long)(i[0] + a.g.nbc[0]) + a.g.sy * (i[1] + a.g.nbc[1]) + a.g.sz * (i[2] + a.g.nbc[2]);
  double *T = a.T;
  switch (a.type) {
    case PION_BC_PERIODIC: {
      // periodic_boundaries.cpp:42-50: NG(axis) cells back onto the grid
      const long s = pos ? c - st * a.g.ng[ax] : c + st * a.g.ng[ax];
      for (int v = 0; v < a.nvar; v++) T[v * nc + c] = T[v * nc + s];
      break;
    }
    case PION_BC_OUTFLOW:
    case PION_BC_ONEWAY_OUT:
    case PION_BC_AXISYMMETRIC:
    case PION_BC_JETREFLECT:
    case PION_BC_REFLECTING: {
      // all ghost layers copy the FIRST on-grid cell of the row (outflow_boundaries.cpp:50-59)
      const long s = pos ? c - st * depth : c + st * depth;
      if (a.type == PION_BC_REFLECTING || a.type == PION_BC_AXISYMMETRIC || a.type == PION_BC_JETREFLECT) {
        // reflecting_boundaries.cpp:34-73,131-153: normal velocity (and normal B) flip sign;
        // axisymmetric_boundaries.cpp:34-52,98-137 (R = 0 axis): the radial and the theta components do
        const bool mhd = (a.eqntype == EQMHD || a.eqntype == EQGLM);
        const bool axi = (a.type == PION_BC_AXISYMMETRIC);
        for (int v = 0; v < a.nvar; v++) {
          double r = 1.0;
          if (axi) {
            if (v == 3 || v == 4) r = -1.0;
            if (mhd && (v == 6 || v == 7)) r = -1.0;
          }
          else if (a.type == PION_BC_JETREFLECT) {
            // jetreflect_boundaries.cpp:32-62: v_n and the two tangential field components
            if (v == 2 + ax) r = -1.0;
            if (mhd && v >= 5 && v <= 7 && v != 5 + ax) r = -1.0;
          }
          else {
            if (v == 2 + ax) r = -1.0;
            if (mhd && v == 5 + ax) r = -1.0;
          }
          T[v * nc + c] = T[v * nc + s] * r;
        }
      }
      else {
        for (int v = 0; v < a.nvar; v++) T[v * nc + c] = T[v * nc + s];
        if (a.type == PION_BC_ONEWAY_OUT) {
          // oneway_out_boundaries.cpp:75-138
          const int vn = 2 + ax;
          const double sg = pos ? 1.0 : -1.0;
          const double x = T[vn * nc + c] * sg;
          T[vn * nc + c] = sg * ((0.0 < x) ? x : 0.0);
        }
        if (a.eqntype == EQGLM) {
          // GLM_NEGATIVE_BOUNDARY (boundaries.h:21, outflow_boundaries.cpp:140-152):
          // psi_ghost(layer d) = -psi(on-grid cell d)
          const long gsrc = pos ? s - st * (depth - 1) : s + st * (depth - 1);
          T[8 * nc + c] = -T[8 * nc + gsrc];
        }
      }
      break;
    }
    case PION_BC_INFLOW:
    case PION_BC_FIXED:
      for (int v = 0; v < a.nvar; v++) T[v * nc + c] = a.refval[v];
      break;
    case PION_BC_DMACH: {
      // double_Mach_ref_boundaries.cpp:168-204
      const double x = a.g.xmin[0] + (2 * i[0] + 1) * (0.5 * a.g.dx);
      const double y = a.g.xmin[1] + (2 * i[1] + 1) * (0.5 * a.g.dx);
      const double bpos = a.dmr_a0 + 1.0 / 6.0 + y / a.dmr_t3;
      if (x <= bpos) {
        T[0 * nc + c] = 8.0;
        T[1 * nc + c] = 116.5;
        T[2 * nc + c] = 7.14470958;
        T[3 * nc + c] = -4.125;
        T[4 * nc + c] = 0.0;
        for (int v = a.nvar - a.ntracer; v < a.nvar; v++) T[v * nc + c] = 1.0;
      }
      else {
        for (int v = 0; v < a.nvar; v++) T[v * nc + c] = a.refval[v];
      }
      break;
    }
    default:
      break;
  }
}

// internal DMR2 boundary: y<0 ghost cells above on-grid columns with x<=1/6 get a fixed state
__global__ void k_bc_dmr2(const BCArgs a, const int ncols)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int nb = a.g.nbc[1];
  if (t >= ncols * nb) return;
  const int ix = t % ncols, iy = -1 - (t / ncols);
  const long nc = a.g.ncell;
  const long c = (long)(ix + a.g.nbc[0]) + a.g.sy * (iy + a.g.nbc[1]);
  for (int v = 0; v < a.nvar; v++) a.T[v * nc + c] = a.refval[v];
}

__global__ void k_wind(double *T, const long *idx, const double *states, const long n, const int nvar,
                       const long nc)
{
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const long c = idx[t];
  for (int v = 0; v < nvar; v++) T[v * nc + c] = states[t * nvar + v];
}

// halo planes: buffer layout [nvar][nbc][ny_all][nx_all]
__global__ void k_halo(double *A, double *buf, const GridDesc g, const int nvar, const int face, const int pack)
{
  const long plane = (long)g.nga[0] * g.nga[1];
  const long per = plane * g.nbc[2];
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per * nvar) return;
  const int v = (int)(t / per);
  const long r = t % per;
  const int k = (int)(r / plane);
  const long xy = r % plane;
  int izall;
  if (pack) izall = (face == 4) ? g.nbc[2] + k : g.ng[2] + k;              // on-grid planes next to the face
  else izall = (face == 4) ? k : g.nbc[2] + g.ng[2] + k;                   // ghost planes of the face
  const long c = xy + plane * izall;
  if (pack) buf[t] = A[v * g.ncell + c];
  else A[v * g.ncell + c] = buf[t];
}

struct Handle {
  pion_gpu_config cfg;
  GridDesc g;
  int device = 0;
  int ncu = 0;            // compute units of the device (launch shaping)
  hipStream_t stream = 0;
  hipStream_t comm_stream = 0;     // pack/unpack of the z halo (0: the compute stream)
  hipEvent_t ev_packed_src = nullptr, ev_unpacked = nullptr;
  hipStream_t bstream = 0;         // the z-boundary strips of a split stage (two-stream mode): beside the interior part
  hipEvent_t ev_pre = nullptr, ev_bdone = nullptr;
  bool ev_pre_valid = false;
  bool concurrent_strips = true;   // PION_CONCURRENT_STRIPS=0: strips after the interior part on the compute stream
  bool ev_unpacked_valid = false;
  double *dP = nullptr, *dPh = nullptr;
  bool own_state = true;
  uint8_t *dflags = nullptr, *dhll = nullptr;
  double *deta = nullptr;
  double *dsphvol = nullptr;   // spherical 1-D: shell volumes/(4 pi) per cell
  int *derr = nullptr;
  unsigned long long *ddt = nullptr;   // [0]=min t_dyn, [1]=min t_mp (bit patterns)
  unsigned long long *ddt_init = nullptr;  // {1e100, 1e99} on the device: reset source (no host buffer in flight)
  double *hdt = nullptr;               // pinned host staging of {t_dyn, t_mp, error word} (pion_gpu_dt_request)
  hipEvent_t ev_dt = nullptr;
  bool dt_requested = false;
  std::vector<uint8_t> hflags;
  // boundary state
  double refval[6][PION_MAX_NVAR];
  int dmr2_cols = 0;
  long nwind = 0;
  long njet = 0;          // jet inflow cells (XN ghosts), one state for all
  long *djet_idx = nullptr;
  double *djet_state = nullptr;
  long *dwind_idx = nullptr;
  double *dwind_state = nullptr;
  // cooling
  CoolDev cool;
  double *dcoolT = nullptr, *dcooltab = nullptr, *dcoolslope = nullptr;
  bool have_tables = false;
  // solver state
  double glm_chyp = 0.0, glm_cr = 0.0;
  double refvec_avg[PION_MAX_NVAR];
  bool ph_valid = false;  // dPh holds a genuine half-step state
  bool dt_cached = false; // ddt holds the time-step minima of the current P (left by the last full stage)
  std::string err;
  // timing
  bool timing = false;
  std::vector<hipEvent_t> ev[4];
  double Mu_tot_over_kB = 0.0;
  int use_march = 3, zchunk = 0, rows = 0;  // zchunk 0: chosen per launch; rows 0: chosen per instance (stage_rows2_rows)
  int rows1 = 0;                            // rows of the first-order stage (PION_ROWS1)
  const double *xghost_fresh = nullptr;   // array whose x ghosts (periodic x) the last stage kernel wrote itself
  int zslope_lds = 1;     // k_stage_rows2: carry the z slope in LDS (default; PION_ZSLOPE_LDS=0: rebuild it from plane k-1, R = 4)
  double *ddE = nullptr;  // cooling source per cell (k_cooling_dE -> k_stage_rows2)
  bool fuse_dt = true;    // PION_FUSE_DT=0: always run k_dt (A/B)
  bool uneven_chunks = true;   // PION_UNEVEN_CHUNKS=0: equal plane chunks (A/B)
  bool dt_mp_pending = false;  // k_dt_mp owed after the two streams of a split stage have joined
  bool split_dt_mp = true;     // PION_SPLIT_DT_MP=0: cooling time inside the stage kernel's fused reduction (A/B)
  bool fuse_bc = true;    // PION_FUSE_BC=0: periodic faces one launch per face (A/B)
};

#define HCHECK(h, call)                                                            \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) {                                                        \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                \
      return PION_GPU_EDEVICE;                                                     \
    }                                                                              \
  } while (0)

// Every entry point selects the handle's device first: the current device is per-thread state, and
// distinct handles may be driven from distinct host threads (or interleaved on one thread).
static inline Handle *use(void *handle)
{
  Handle *h = (Handle *)handle;
  if (h) (void)hipSetDevice(h->device);
  return h;
}

// Two-stream mode: everything on the compute stream that touches the z ghost planes must run after
// the last unpack on the comm stream.  One wait is enough, later work is ordered behind it.
int order_after_unpack(Handle *h)
{
  if (h->comm_stream && h->comm_stream != h->stream && h->ev_unpacked_valid) {
    HCHECK(h, hipStreamWaitEvent(h->stream, h->ev_unpacked, 0));
    h->ev_unpacked_valid = false;
  }
  return 0;
}

long cell_id(const GridDesc &g, int ix, int iy, int iz)
{
  return (long)(ix + g.nbc[0]) + g.sy * (iy + g.nbc[1]) + g.sz * (iz + g.nbc[2]);
}

void time_begin(Handle *h, int slot)
{
  if (!h->timing) return;
  hipEvent_t e;
  hipEventCreate(&e);
  hipEventRecord(e, h->stream);
  h->ev[slot].push_back(e);
}
void time_end(Handle *h, int slot) { time_begin(h, slot); }

FluxCtx make_fluxctx(const Handle *h, double fv_dt)
{
  FluxCtx fc;
  fc.gamma = h->cfg.gamma;
  fc.dx = h->cfg.dx;
  fc.fv_dt = fv_dt;
  fc.etav = h->cfg.etav;
  fc.chyp = h->glm_chyp;
  fc.min_temp = h->cfg.min_temp;
  fc.refRO = h->refvec_avg[0];
  fc.refPG = h->refvec_avg[1];
  fc.refV = h->refvec_avg[2];
  fc.refB = h->refvec_avg[5];
  fc.gndim = h->cfg.ndim;
  fc.artvisc = h->cfg.artvisc;
  fc.mp.present = (h->cfg.cooling != 0);
  fc.mp.Mu_tot_over_kB = h->Mu_tot_over_kB;
  return fc;
}

int check_errword(Handle *h)
{
  int e = 0;
  HCHECK(h, hipMemcpyAsync(&e, h->derr, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HCHECK(h, hipStreamSynchronize(h->stream));
  if (e) {
    char b[400];
    snprintf(b, sizeof b, "device physics error word 0x%x:%s%s%s%s%s", e,
             (e & ERR_NEG_DENSITY) ? " negative density (reference: rep.error -> exit)" : "",
             (e & ERR_RIEMANN_INPUT) ? " density/pressure too small in Riemann solver" : "",
             (e & ERR_COOLING) ? " cooling integration failed" : "", (e & ERR_BAD_DT) ? " invalid cell timestep" : "",
             (e & ERR_MHD_RIEMANN) ? " linear MHD Riemann solver: bad wave speeds (reference: rep.error -> exit)" : "");
    h->err = b;
    int z = 0;
    hipMemcpyAsync(h->derr, &z, sizeof(int), hipMemcpyHostToDevice, h->stream);
    return PION_GPU_EPHYSICS;
  }
  return 0;
}

// get_mp_timescales_no_radiation (calc_timestep.cpp:445-459): EP.MP_timestep_limit 1, 2, 3 ask
// mp_only_cooling::timescales for the cooling time (tc = true); 4 (recombination time only) gets 1e99
// from it (mp_only_cooling.cpp:338), i.e. no limit; anything else is fatal there (EINVAL in create).
static inline bool mp_dt_limited(const pion_gpu_config &cfg)
{
  return cfg.cooling != 0 && cfg.mp_timestep_limit >= 1 && cfg.mp_timestep_limit <= 3;
}

// device scratch of the test seams: freed on every return path
struct DevBuf {
  double *p = nullptr;
  ~DevBuf()
  {
    if (p) (void)hipFree(p);
  }
};

}  // namespace

extern "C" {

int pion_gpu_create(const pion_gpu_config *cfg, int device, void **handle)
{
  if (!cfg || !handle) return PION_GPU_EINVAL;
  if (cfg->ndim < 1 || cfg->ndim > 3 || cfg->nvar > PION_MAX_NVAR) return PION_GPU_EINVAL;
  // Cartesian, or cylindrical (z,R) axisymmetry in 2-D (the only cylindrical case the reference's solver
  // classes accept: solver_eqn_hydro_adi.cpp:540-545, solver_eqn_mhd_adi.cpp:985-990)
  // or spherical symmetry in 1-D, hydro only (sph_FV_solver_Hydro_Euler, solver_eqn_hydro_adi.cpp:620-640)
  if (!(cfg->coord_sys == 1 || (cfg->coord_sys == 2 && cfg->ndim == 2)
        || (cfg->coord_sys == 3 && cfg->ndim == 1 && cfg->eqntype == PION_EQEUL)))
    return PION_GPU_EINVAL;
  for (int d = 0; d < 2 * cfg->ndim; d++)
    if (cfg->bc_type[d] == PION_BC_AXISYMMETRIC && !(cfg->coord_sys == 2 && d == 2)) return PION_GPU_EINVAL;
  const int base = (cfg->eqntype == PION_EQEUL) ? 5 : (cfg->eqntype == PION_EQMHD ? 8 : (cfg->eqntype == PION_EQGLM ? 9 : -1));
  if (base < 0 || cfg->nvar != base + cfg->ntracer || cfg->ntracer > PION_MAX_NTR) return PION_GPU_EINVAL;
  if (cfg->sp_ooa == 2 && cfg->nbc < 2) return PION_GPU_EINVAL;
  if (cfg->nbc < 1) return PION_GPU_EINVAL;
  if (cfg->eqntype == PION_EQEUL) {
    if (cfg->solver == 7 || cfg->solver < 0 || cfg->solver > 8) return PION_GPU_EINVAL;
  }
  // MHD: LF, FKJ98 linear, Roe, HLLD, HLL (exact/hybrid are fatal in riemannMHD.cpp:176-181)
  else if (!(cfg->solver == 0 || cfg->solver == 1 || cfg->solver == 4 || cfg->solver == 7 || cfg->solver == 8))
    return PION_GPU_EINVAL;
  if (cfg->cooling != 0 && cfg->cooling != PION_COOL_WSS09_CIE_LINE_HEAT_COOL) return PION_GPU_EINVAL;
  if (cfg->mp_timestep_limit < 0 || cfg->mp_timestep_limit > 4) return PION_GPU_EINVAL;  // calc_timestep.cpp:457

  Handle *h = new Handle;
  h->cfg = *cfg;
  if (const char *e = getenv("PION_STAGE_KERNEL"))
    h->use_march = (strcmp(e, "cell") == 0) ? 0 : 3;
  if (const char *e = getenv("PION_ZSLOPE_LDS")) h->zslope_lds = (atoi(e) != 0);
  if (const char *e = getenv("PION_CONCURRENT_STRIPS")) h->concurrent_strips = (atoi(e) != 0);
  if (const char *e = getenv("PION_FUSE_DT")) h->fuse_dt = (atoi(e) != 0);
  if (const char *e = getenv("PION_FUSE_BC")) h->fuse_bc = (atoi(e) != 0);
  if (const char *e = getenv("PION_UNEVEN_CHUNKS")) h->uneven_chunks = (atoi(e) != 0);
  if (const char *e = getenv("PION_SPLIT_DT_MP")) h->split_dt_mp = (atoi(e) != 0);
  if (const char *e = getenv("PION_ROWS")) h->rows = h->rows1 = (atoi(e) >= 1 && atoi(e) <= 64) ? atoi(e) : 0;
  if (const char *e = getenv("PION_ROWS1")) h->rows1 = (atoi(e) >= 1 && atoi(e) <= 8) ? atoi(e) : 0;
  if (const char *e = getenv("PION_ZCHUNK")) h->zchunk = atoi(e) > 0 ? atoi(e) : 0;
  h->device = device;
  if (hipSetDevice(device) != hipSuccess) {
    delete h;
    return PION_GPU_EDEVICE;
  }
  {
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) h->ncu = ncu;
  }
  GridDesc &g = h->g;
  g.ndim = cfg->ndim;
  g.ncell = 1;
  for (int a = 0; a < 3; a++) {
    g.ng[a] = (a < cfg->ndim) ? cfg->ng[a] : 1;
    g.nbc[a] = (a < cfg->ndim) ? cfg->nbc : 0;
    g.nga[a] = g.ng[a] + 2 * g.nbc[a];
    g.ncell *= g.nga[a];
    g.xmin[a] = cfg->xmin[a];
  }
  g.sy = g.nga[0];
  g.sz = (long)g.nga[0] * g.nga[1];
  g.dx = cfg->dx;
  g.cyl = (cfg->coord_sys == 2) ? 1 : ((cfg->coord_sys == 3) ? 2 : 0);
  g.sph_vol = nullptr;
  *handle = h;
  // k_stage_rows2 (3-D, two ghost layers) addresses every array as "uniform base + 32-bit byte offset of the
  // cell": grids of 2^29 cells or more (ghosts included) use the cell-per-thread kernel with 64-bit addresses
  // (2-D Cartesian grids run it without the z part; PION_ROWS_2D=0 puts them back on the cell-per-thread kernel)
  const bool rows3d = (g.ndim == 3 && g.nbc[2] >= 2);
  bool rows2d = (g.ndim == 2 && g.cyl != 2 && g.nbc[1] >= 2 && g.nbc[0] >= 2);   // (cyl == 1: the CYL instance)
  if (const char *e = getenv("PION_ROWS_2D")) rows2d = rows2d && (atoi(e) != 0);
  if (!((rows3d || rows2d) && (unsigned long long)g.ncell * 8ull < (1ull << 32))) h->use_march = 0;

  const size_t nb = sizeof(double) * (size_t)cfg->nvar * g.ncell;
  HCHECK(h, hipMalloc(&h->dP, nb));
  HCHECK(h, hipMalloc(&h->dPh, nb));
  HCHECK(h, hipMemset(h->dP, 0, nb));
  HCHECK(h, hipMemset(h->dPh, 0, nb));
  HCHECK(h, hipMalloc(&h->dflags, g.ncell));
  if (g.cyl == 2) {
    // VectorOps_Sph::DivStateVectorComponent: rc = (pow(rp,3.0) - pow(rn,3.0))/3.0 with the host's libm
    std::vector<double> vol(g.nga[0]);
    for (int i = 0; i < g.nga[0]; i++) {
      double rc = g.xmin[0] + (2 * (i - g.nbc[0]) + 1) * (0.5 * g.dx);
      const double rp = rc + 0.5 * g.dx;
      const double rn = rp - g.dx;
      vol[i] = (pow(rp, 3.0) - pow(rn, 3.0)) / 3.0;
    }
    HCHECK(h, hipMalloc(&h->dsphvol, sizeof(double) * vol.size()));
    HCHECK(h, hipMemcpy(h->dsphvol, vol.data(), sizeof(double) * vol.size(), hipMemcpyHostToDevice));
    g.sph_vol = h->dsphvol;
  }
  HCHECK(h, hipMalloc(&h->derr, 64));
  HCHECK(h, hipMemset(h->derr, 0, 64));
  HCHECK(h, hipMalloc(&h->ddt, 2 * sizeof(unsigned long long)));
  HCHECK(h, hipMalloc(&h->ddt_init, 2 * sizeof(unsigned long long)));
  {
    const double init[2] = {1.e100, 1.0e99};
    HCHECK(h, hipMemcpy(h->ddt_init, init, sizeof init, hipMemcpyHostToDevice));
  }
  if (cfg->eqntype != PION_EQEUL && cfg->solver == PION_FLUX_RS_HLLD) {
    HCHECK(h, hipMalloc(&h->dhll, g.ncell));
    HCHECK(h, hipMemset(h->dhll, 0, g.ncell));
  }
  if (cfg->cooling != 0) {
    HCHECK(h, hipMalloc(&h->ddE, sizeof(double) * g.ncell));
    HCHECK(h, hipMemset(h->ddE, 0, sizeof(double) * g.ncell));
  }
  if (cfg->artvisc == PION_AV_HCORRECTION || cfg->artvisc == PION_AV_HCORR_FKJ98) {
    HCHECK(h, hipMalloc(&h->deta, sizeof(double) * cfg->ndim * g.ncell));
    HCHECK(h, hipMemset(h->deta, 0, sizeof(double) * cfg->ndim * g.ncell));
  }

  // cell flags (uniform_grid.cpp:343-356,516-546; periodic ghosts are isdomain,
  // periodic_boundaries.cpp:35-36; everything else off-grid is not)
  h->hflags.assign(g.ncell, 0);
  for (long c = 0; c < g.ncell; c++) {
    int i[3];
    i[0] = (int)(c % g.nga[0]) - g.nbc[0];
    i[1] = (int)((c / g.nga[0]) % g.nga[1]) - g.nbc[1];
    i[2] = (int)(c / g.sz) - g.nbc[2];
    bool on = true;
    bool all_periodic_offgrid = true;
    for (int a = 0; a < cfg->ndim; a++) {
      if (i[a] < 0) {
        on = false;
        if (cfg->bc_type[2 * a] != PION_BC_PERIODIC) all_periodic_offgrid = false;
      }
      else if (i[a] >= g.ng[a]) {
        on = false;
        if (cfg->bc_type[2 * a + 1] != PION_BC_PERIODIC) all_periodic_offgrid = false;
      }
    }
    uint8_t f = PION_CELL_ISLEAF | PION_CELL_TIMESTEP;
    if (on) f |= PION_CELL_ISGD | PION_CELL_ISDOMAIN;
    else {
      f |= PION_CELL_ISBD;
      (void)all_periodic_offgrid;  // ghost isdomain never matters on the device: ghosts are not updated
    }
    h->hflags[c] = f;
  }
  HCHECK(h, hipMemcpy(h->dflags, h->hflags.data(), g.ncell, hipMemcpyHostToDevice));

  // microphysics constants (mp_only_cooling.cpp:81-95,140-146; constants.h:53,64)
  const double m_p = 1.672621898e-24, kB = 1.38064852e-16;
  const double Mu = 1.40 * m_p, Mu_tot = 0.609 * m_p, Mu_elec = 1.167 * m_p;
  h->Mu_tot_over_kB = Mu_tot / kB;
  memset(&h->cool, 0, sizeof h->cool);
  h->cool.inv_Mu2 = 1.0 / (Mu * Mu);
  h->cool.inv_Mu2_elec_H = 1.0 / (Mu_elec * Mu);
  h->cool.Mu_tot_over_kB = h->Mu_tot_over_kB;
  h->cool.MinT_allowed = cfg->min_temp;
  h->cool.MaxT_allowed = cfg->max_temp;
  if (h->cool.MinT_allowed < 1.0 || h->cool.MinT_allowed > 1.0e6) h->cool.MinT_allowed = 1.0;
  if (h->cool.MaxT_allowed < 1.0e2 || h->cool.MaxT_allowed > 3.0e10) h->cool.MaxT_allowed = 1.0e8;

  // eq_refvec after SetAvgState (eqns_hydro_adiabatic.cpp:437-453); only the Euler Riemann
  // solvers (riemann.cpp) read it
  for (int v = 0; v < PION_MAX_NVAR; v++) h->refvec_avg[v] = cfg->refvec[v];
  if (cfg->eqntype == PION_EQEUL) {
    const double refvel = sqrt(cfg->gamma * cfg->refvec[1] / cfg->refvec[0]);
    h->refvec_avg[2] = h->refvec_avg[3] = h->refvec_avg[4] = 0.1 * refvel;
  }
  else {
    // eqns_mhd_ideal::SetAvgState (eqns_mhd_adiabatic.cpp:501-544): fast speed of the reference
    // state with the field rotated into the x-y plane's x axis; velocities <- 0.1 c_f, fields <- |B|
    double *rv = h->refvec_avg;
    const double gam = cfg->gamma;
    auto cfast = [&](const double *p) {
      const double ch = sqrt(gam * p[1] / p[0]);
      const double t1 = ch * ch + (p[5] * p[5] + p[6] * p[6] + p[7] * p[7]) / p[0];
      double t2 = 4. * ch * ch * p[5] * p[5] / p[0];
      t2 = std::max(5.e-16, t1 * t1 - t2);
      return sqrt((t1 + sqrt(t2)) / 2.);
    };
    auto rotate_xy = [&](double *v, double theta) {
      const double ct = cos(theta), st = sin(theta);
      double a = v[2] * ct - v[3] * st, b = v[2] * st + v[3] * ct;
      v[2] = a;
      v[3] = b;
      a = v[5] * ct - v[6] * st;
      b = v[5] * st + v[6] * ct;
      v[5] = a;
      v[6] = b;
    };
    double angle = rv[6] * rv[6] + rv[5] * rv[5], refvel;
    if (angle > 10. * 5.e-16) {
      angle = M_PI / 2. - asin(rv[6] / sqrt(angle));
      if (rv[5] < 0) angle = -angle;
      rotate_xy(rv, angle);
      refvel = cfast(rv);
      rotate_xy(rv, -angle);
    }
    else refvel = cfast(rv);
    const double refB = sqrt(rv[5] * rv[5] + rv[6] * rv[6] + rv[7] * rv[7]);
    rv[2] = rv[3] = rv[4] = 0.1 * refvel;
    rv[5] = rv[6] = rv[7] = refB;
  }
  for (int d = 0; d < 6; d++)
    for (int v = 0; v < PION_MAX_NVAR; v++) h->refval[d][v] = 0.0;

  // DMR2: on-grid columns with x <= 1/6
  if (cfg->bc_dmach2) {
    int n = 0;
    for (int ix = 0; ix < g.ng[0]; ix++) {
      const double x = g.xmin[0] + (2 * ix + 1) * (0.5 * g.dx);
      if (x <= 1. / 6.) n++;
      else break;
    }
    h->dmr2_cols = n;
  }
  return PION_GPU_OK;
}

void pion_gpu_destroy(void *handle)
{
  Handle *h = use(handle);
  if (!h) return;
  hipSetDevice(h->device);
  hipDeviceSynchronize();
  if (h->own_state) {
    hipFree(h->dP);
    hipFree(h->dPh);
  }
  hipFree(h->dflags);
  hipFree(h->dhll);
  hipFree(h->ddE);
  if (h->bstream) hipStreamDestroy(h->bstream);
  if (h->ev_pre) hipEventDestroy(h->ev_pre);
  if (h->ev_bdone) hipEventDestroy(h->ev_bdone);
  if (h->hdt) hipHostFree(h->hdt);
  if (h->ev_dt) hipEventDestroy(h->ev_dt);
  hipFree(h->deta);
  hipFree(h->dsphvol);
  hipFree(h->derr);
  hipFree(h->ddt);
  hipFree(h->ddt_init);
  hipFree(h->dwind_idx);
  hipFree(h->dwind_state);
  hipFree(h->djet_idx);
  hipFree(h->djet_state);
  hipFree(h->dcoolT);
  hipFree(h->dcooltab);
  hipFree(h->dcoolslope);
  for (int s = 0; s < 4; s++)
    for (hipEvent_t e : h->ev[s]) hipEventDestroy(e);
  if (h->ev_packed_src) hipEventDestroy(h->ev_packed_src);
  if (h->ev_unpacked) hipEventDestroy(h->ev_unpacked);
  delete h;
}

int pion_gpu_last_error(void *handle, char *buf, int len)
{
  Handle *h = use(handle);
  if (!h || !buf || len <= 0) return PION_GPU_EINVAL;
  snprintf(buf, len, "%s", h->err.c_str());
  return 0;
}

long pion_gpu_ncell_all(void *handle) { return ((Handle *)handle)->g.ncell; }
int pion_gpu_ng_all(void *handle, int axis) { return ((Handle *)handle)->g.nga[axis]; }

int pion_gpu_upload(void *handle, const double *P_soa)
{
  Handle *h = use(handle);
  h->xghost_fresh = nullptr;
  const size_t nb = sizeof(double) * (size_t)h->cfg.nvar * h->g.ncell;
  HCHECK(h, hipMemcpyAsync(h->dP, P_soa, nb, hipMemcpyHostToDevice, h->stream));
  HCHECK(h, hipMemcpyAsync(h->dPh, h->dP, nb, hipMemcpyDeviceToDevice, h->stream));
  HCHECK(h, hipStreamSynchronize(h->stream));
  h->ph_valid = false;
  h->dt_cached = false;
  h->dt_requested = false;   // a read-back requested for the previous state is void
  h->dt_mp_pending = false;
  return 0;
}

int pion_gpu_download(void *handle, int which, double *P_soa)
{
  Handle *h = use(handle);
  const size_t nb = sizeof(double) * (size_t)h->cfg.nvar * h->g.ncell;
  // after a full step the reference has Ph == P everywhere (time_integrator.cpp:938-939)
  const double *src = (which == 1 && h->ph_valid) ? h->dPh : h->dP;
  if (int rc = order_after_unpack(h)) return rc;
  HCHECK(h, hipMemcpyAsync(P_soa, src, nb, hipMemcpyDeviceToHost, h->stream));
  HCHECK(h, hipStreamSynchronize(h->stream));
  return check_errword(h);
}

int pion_gpu_bind_device_state(void *handle, void *dP, void *dPh)
{
  Handle *h = use(handle);
  h->xghost_fresh = nullptr;
  if (!dP || !dPh) return PION_GPU_EINVAL;
  if (h->own_state) {
    hipFree(h->dP);
    hipFree(h->dPh);
  }
  h->own_state = false;
  h->dP = (double *)dP;
  h->dPh = (double *)dPh;
  h->ph_valid = false;
  h->dt_cached = false;
  h->dt_requested = false;
  return 0;
}
void *pion_gpu_device_ptr(void *handle, int which)
{
  Handle *h = use(handle);
  h->xghost_fresh = nullptr;
  h->dt_cached = false;  // the caller may write through the pointer
  return which == 0 ? (void *)h->dP : (void *)h->dPh;
}
int pion_gpu_set_stream(void *handle, void *stream)
{
  ((Handle *)handle)->stream = (hipStream_t)stream;
  return 0;
}
int pion_gpu_set_comm_stream(void *handle, void *stream)
{
  Handle *h = use(handle);
  h->comm_stream = (hipStream_t)stream;
  h->ev_unpacked_valid = false;
  return 0;
}
int pion_gpu_synchronize(void *handle)
{
  Handle *h = use(handle);
  HCHECK(h, hipStreamSynchronize(h->stream));
  if (h->comm_stream && h->comm_stream != h->stream) HCHECK(h, hipStreamSynchronize(h->comm_stream));
  if (h->bstream) HCHECK(h, hipStreamSynchronize(h->bstream));
  return 0;
}

int pion_gpu_set_wind_cells(void *handle, long n, const long *idx, const double *states)
{
  Handle *h = use(handle);
  h->dt_cached = false;   // the ISBD flags decide which cells enter the time-step reduction
  hipFree(h->dwind_idx);
  hipFree(h->dwind_state);
  h->dwind_idx = nullptr;
  h->dwind_state = nullptr;
  h->nwind = n;
  if (n > 0) {
    HCHECK(h, hipMalloc(&h->dwind_idx, sizeof(long) * n));
    HCHECK(h, hipMalloc(&h->dwind_state, sizeof(double) * n * h->cfg.nvar));
    HCHECK(h, hipMemcpy(h->dwind_idx, idx, sizeof(long) * n, hipMemcpyHostToDevice));
    HCHECK(h, hipMemcpy(h->dwind_state, states, sizeof(double) * n * h->cfg.nvar, hipMemcpyHostToDevice));
    for (long k = 0; k < n; k++) {
      if (idx[k] < 0 || idx[k] >= h->g.ncell) return PION_GPU_EINVAL;
      h->hflags[idx[k]] |= PION_CELL_ISBD;  // stellar_wind_BC.cpp:277-278
      h->hflags[idx[k]] &= ~PION_CELL_ISDOMAIN;
    }
    HCHECK(h, hipMemcpy(h->dflags, h->hflags.data(), h->g.ncell, hipMemcpyHostToDevice));
  }
  return 0;
}

int pion_gpu_set_jet(void *handle, int jetradius, const double *jetstate)
{
  Handle *h = use(handle);
  h->dt_cached = false;   // (cell flags change)
  const pion_gpu_config &cfg = h->cfg;
  const GridDesc &g = h->g;
  const bool cart3d = (cfg.ndim == 3 && cfg.coord_sys == 1 && cfg.eqntype == PION_EQEUL);
  const bool cyl2d = (cfg.ndim == 2 && cfg.coord_sys == 2);
  if ((!cart3d && !cyl2d) || !jetstate) {
    h->err = "jet boundary: 3-D Cartesian Euler or 2-D cylindrical only (jet_boundaries.cpp:88-91,203-206)";
    return PION_GPU_EINVAL;
  }
  std::vector<long> idx;
  if (cart3d) {
    // BC_assign_JETBC, 3-D Cartesian (jet_boundaries.cpp:170-201)
    const double jr = jetradius * g.dx;
    for (int iz = 0; iz < g.ng[2]; iz++)
      for (int iy = 0; iy < g.ng[1]; iy++) {
        const double y = g.xmin[1] + (2 * iy + 1) * (0.5 * g.dx), z = g.xmin[2] + (2 * iz + 1) * (0.5 * g.dx);
        if (sqrt(y * y + z * z) <= jr)
          for (int k = 1; k <= g.nbc[0]; k++) idx.push_back(cell_id(g, -k, iy, iz));
      }
  }
  else {
    // 2-D axisymmetric (:96-168): the first jetradius rows above the axis; the profile written at
    // assignment does not survive the first update (:212-262), so the uniform state is all there is
    if (jetradius > g.ng[1]) {
      h->err = "Not enough cells for jet";
      return PION_GPU_EINVAL;
    }
    for (int iy = 0; iy < jetradius; iy++)
      for (int k = 1; k <= g.nbc[0]; k++) idx.push_back(cell_id(g, -k, iy, 0));
  }
  // refval (jet_boundaries.cpp:60-93): 2-D MHD keeps B along the axis and the toroidal component
  std::vector<double> rv(jetstate, jetstate + cfg.nvar);
  if (cfg.eqntype != PION_EQEUL) {
    rv[5] = jetstate[5];
    rv[6] = 0.0;
    rv[7] = jetstate[6];
  }
  hipFree(h->djet_idx);
  hipFree(h->djet_state);
  h->djet_idx = nullptr;
  h->djet_state = nullptr;
  h->njet = (long)idx.size();
  // k_wind takes one state per cell
  std::vector<double> st((size_t)h->njet * cfg.nvar);
  for (long k = 0; k < h->njet; k++)
    for (int v = 0; v < cfg.nvar; v++) st[(size_t)k * cfg.nvar + v] = rv[v];
  if (h->njet > 0) {
    HCHECK(h, hipMalloc(&h->djet_idx, sizeof(long) * h->njet));
    HCHECK(h, hipMalloc(&h->djet_state, sizeof(double) * st.size()));
    HCHECK(h, hipMemcpy(h->djet_idx, idx.data(), sizeof(long) * h->njet, hipMemcpyHostToDevice));
    HCHECK(h, hipMemcpy(h->djet_state, st.data(), sizeof(double) * st.size(), hipMemcpyHostToDevice));
  }
  return 0;
}

int pion_gpu_set_cooling_tables(void *handle, int nT, const double *T, const double *tabs, const double *slopes)
{
  Handle *h = use(handle);
  h->dt_cached = false;   // t_mp depends on the tables
  if (nT < 2 || nT > PION_COOL_NT_MAX) {
    // (k_cooling_dE keeps the tables in LDS: 11 x PION_COOL_NT_MAX doubles; mp_only_cooling builds 200 points)
    h->err = "cooling tables: 2 <= nT <= 256 required";
    return PION_GPU_EINVAL;
  }
  hipFree(h->dcoolT);
  hipFree(h->dcooltab);
  hipFree(h->dcoolslope);
  HCHECK(h, hipMalloc(&h->dcoolT, sizeof(double) * nT));
  HCHECK(h, hipMalloc(&h->dcooltab, sizeof(double) * 5 * nT));
  HCHECK(h, hipMalloc(&h->dcoolslope, sizeof(double) * 5 * nT));
  HCHECK(h, hipMemcpy(h->dcoolT, T, sizeof(double) * nT, hipMemcpyHostToDevice));
  HCHECK(h, hipMemcpy(h->dcooltab, tabs, sizeof(double) * 5 * nT, hipMemcpyHostToDevice));
  HCHECK(h, hipMemcpy(h->dcoolslope, slopes, sizeof(double) * 5 * nT, hipMemcpyHostToDevice));
  // log-spaced grid?  (mp_only_cooling's is: T_i = 10^(log10 Tmin + i dlogT), mp_only_cooling.cpp:533-537.)  Then the
  // device finds the table interval from a single-precision logarithm instead of bisecting; the guess only has to
  // land within a few entries of the truth -- it is corrected against the table -- so a loose check suffices.
  h->cool.lg0 = 0.0f;
  h->cool.inv_dlg = 0.0f;
  if (T[0] > 0.0 && T[nT - 1] > T[0]) {
    const double l0 = log2(T[0]), inv = (nT - 1) / (log2(T[nT - 1]) - l0);
    bool ok = true;
    for (int i = 0; i < nT && ok; i++) {
      if (!(T[i] > 0.0) || (i > 0 && !(T[i] > T[i - 1]))) ok = false;
      else if (fabs((log2(T[i]) - l0) * inv - i) > 0.25) ok = false;
    }
    if (ok) {
      h->cool.lg0 = (float)l0;
      h->cool.inv_dlg = (float)inv;
    }
  }
  if (const char *e = getenv("PION_COOL_BISECT")) {
    if (atoi(e) != 0) h->cool.inv_dlg = 0.0f;   // A/B and cross-check: the reference's bisection
  }
  h->cool.NT = nT;
  h->cool.T = h->dcoolT;
  h->cool.tab = h->dcooltab;
  h->cool.slope = h->dcoolslope;
  h->have_tables = true;
  return 0;
}

int pion_gpu_update_bcs(void *handle, double simtime, int cstep, int maxstep, int assign)
{
  Handle *h = use(handle);
  const pion_gpu_config &cfg = h->cfg;
  const GridDesc &g = h->g;
  const bool full = (cstep == maxstep);
  // after a partial step only Ph's ghosts are refreshed, after the full step P's (and Ph=P)
  double *T = full ? h->dP : h->dPh;
  time_begin(h, 2);

  // TimeUpdateInternalBCs: stellar wind only (assign_update_bcs.cpp:134-183)
  if (h->nwind > 0) {
    hipLaunchKernelGGL(k_wind, dim3((unsigned)((h->nwind + 255) / 256)), dim3(256), 0, h->stream, T, h->dwind_idx,
                       h->dwind_state, h->nwind, cfg.nvar, g.ncell);
  }
  // every face periodic (z possibly handed to the neighbour ranks): one launch fills all ghosts
  bool all_periodic = (h->nwind == 0 && !cfg.bc_dmach2 && h->fuse_bc);
  for (int d = 0; d < 2 * cfg.ndim && all_periodic; d++) {
    const bool zface = (d >= 4);
    if (!(cfg.bc_type[d] == PION_BC_PERIODIC || (zface && cfg.bc_type[d] == PION_BC_SLAB))) all_periodic = false;
  }
  if (all_periodic && cfg.ndim == 3 && cfg.bc_type[4] != cfg.bc_type[5]) all_periodic = false;
  if (all_periodic) {
    const int zwrap = (cfg.ndim == 3 && cfg.bc_type[4] == PION_BC_PERIODIC) ? 1 : 0;
    // x ghosts of the on-grid rows: already in place when the stage kernel that wrote T also wrote them
    const int skipx = (h->xghost_fresh == T) ? 1 : 0;
    const long n = (zwrap ? (long)2 * g.nbc[2] * g.nga[0] * g.nga[1] : 0) + (long)g.ng[2] * 2 * g.nbc[1] * g.nga[0]
                   + (skipx ? 0 : (long)g.ng[2] * g.ng[1] * 2 * g.nbc[0]);
    hipLaunchKernelGGL(k_bc_periodic_all, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, T, g, cfg.nvar,
                       zwrap, skipx);
  }
  h->xghost_fresh = nullptr;
  // any other mix of face types, once the boundaries are assigned: ONE launch for all external faces and the
  // internal DMR2 boundary (k_bc_all); the assignment itself (inflow / fixed states are captured face by face,
  // after the lower faces were filled) keeps the per-face sequence below
  const bool one_launch = (!all_periodic && !assign && h->fuse_bc);
  if (one_launch) {
    BCAllArgs a;
    a.g = g;
    a.T = T;
    a.nvar = cfg.nvar;
    a.eqntype = cfg.eqntype;
    a.ntracer = cfg.ntracer;
    a.ndim = cfg.ndim;
    for (int d = 0; d < 6; d++) {
      a.type[d] = (d < 2 * cfg.ndim) ? cfg.bc_type[d] : 0;
      for (int v = 0; v < PION_MAX_NVAR; v++) a.refval[d][v] = h->refval[d][v];
    }
    a.dmr_a0 = 10.0 * simtime / sin(M_PI / 3.0);
    a.dmr_t3 = tan(M_PI / 3.0);
    a.dmr2_cols = (cfg.bc_dmach2 && h->dmr2_cols > 0) ? h->dmr2_cols : 0;
    for (int v = 0; v < PION_MAX_NVAR; v++) a.dmr2_val[v] = 0.0;
    a.dmr2_val[0] = 8.0;
    a.dmr2_val[1] = 116.5;
    a.dmr2_val[2] = 7.14470958;
    a.dmr2_val[3] = -4.125;
    for (int v = cfg.nvar - cfg.ntracer; v < cfg.nvar; v++) a.dmr2_val[v] = 1.0;
    const long n = ((cfg.ndim == 3) ? (long)2 * g.nbc[2] * g.nga[0] * g.nga[1] : 0)
                   + ((cfg.ndim >= 2) ? (long)g.ng[2] * 2 * g.nbc[1] * g.nga[0] : 0) + (long)g.ng[2] * g.ng[1] * 2 * g.nbc[0];
    hipLaunchKernelGGL(k_bc_all, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, a);
  }
  // TimeUpdateExternalBCs in list order XN,XP,YN,YP,ZN,ZP then DMR2 (assign_update_bcs.cpp:185-252)
  for (int d = 0; d < 2 * cfg.ndim && !all_periodic && !one_launch; d++) {
    const int type = cfg.bc_type[d];
    if (type == 0 || type == PION_BC_SLAB) continue;
    if (assign) {
      // BC_assign_INFLOW / BC_assign_FIXED: the constant state is read from P once, when this
      // boundary is assigned, i.e. after the lower faces have been filled (assign_update_bcs.cpp:58-131;
      // inflow_boundaries.cpp: source of the LAST list cell; fixed_boundaries.cpp:62-76: of the FIRST)
      const int ax = d / 2;
      const bool pos = d & 1;
      if (type == PION_BC_INFLOW || type == PION_BC_FIXED) {
        int i[3] = {0, 0, 0};
        const bool last = (type == PION_BC_INFLOW);
        for (int a = 0; a < 3; a++) {
          if (a == ax) i[a] = pos ? g.ng[a] - 1 : 0;
          else if (a < ax || a >= cfg.ndim) i[a] = last ? g.ng[a] + g.nbc[a] - 1 : -g.nbc[a];
          else i[a] = last ? g.ng[a] - 1 : 0;
        }
        const long c = cell_id(g, i[0], i[1], i[2]);
        HCHECK(h, hipStreamSynchronize(h->stream));
        for (int v = 0; v < cfg.nvar; v++)
          HCHECK(h, hipMemcpy(&h->refval[d][v], h->dP + v * g.ncell + c, sizeof(double), hipMemcpyDeviceToHost));
      }
      else if (type == PION_BC_DMACH) {
        // double_Mach_ref_boundaries.cpp:36-44
        for (int v = 0; v < PION_MAX_NVAR; v++) h->refval[d][v] = 0.0;
        h->refval[d][0] = 1.4;
        h->refval[d][1] = 1.0;
        for (int v = cfg.nvar - cfg.ntracer; v < cfg.nvar; v++) h->refval[d][v] = -1.0;
      }
    }
    BCArgs a;
    a.g = g;
    a.T = T;
    a.nvar = cfg.nvar;
    a.dir = d;
    a.type = type;
    a.eqntype = cfg.eqntype;
    a.ntracer = cfg.ntracer;
    for (int v = 0; v < PION_MAX_NVAR; v++) a.refval[v] = h->refval[d][v];
    a.dmr_a0 = 10.0 * simtime / sin(M_PI / 3.0);
    a.dmr_t3 = tan(M_PI / 3.0);
    const int ax = d / 2;
    long total = g.nbc[ax];
    for (int a2 = 0; a2 < 3; a2++) {
      if (a2 == ax) continue;
      total *= (a2 < ax || a2 >= cfg.ndim) ? g.nga[a2] : g.ng[a2];
    }
    hipLaunchKernelGGL(k_bc_face, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, a);
  }
  if (cfg.bc_dmach2 && h->dmr2_cols > 0 && !one_launch) {
    BCArgs a;
    a.g = g;
    a.T = T;
    a.nvar = cfg.nvar;
    a.dir = -1;
    a.type = PION_BC_DMACH2;
    a.eqntype = cfg.eqntype;
    a.ntracer = cfg.ntracer;
    for (int v = 0; v < PION_MAX_NVAR; v++) a.refval[v] = 0.0;
    a.refval[0] = 8.0;
    a.refval[1] = 116.5;
    a.refval[2] = 7.14470958;
    a.refval[3] = -4.125;
    a.refval[4] = 0.0;
    for (int v = cfg.nvar - cfg.ntracer; v < cfg.nvar; v++) a.refval[v] = 1.0;
    a.dmr_a0 = a.dmr_t3 = 0.0;
    const int n = h->dmr2_cols * g.nbc[1];
    hipLaunchKernelGGL(k_bc_dmr2, dim3((n + 255) / 256), dim3(256), 0, h->stream, a, h->dmr2_cols);
  }
  // internal JETBC, listed after the external boundaries (jet_boundaries.cpp:212-262)
  if (h->njet > 0) {
    hipLaunchKernelGGL(k_wind, dim3((unsigned)((h->njet + 255) / 256)), dim3(256), 0, h->stream, T, h->djet_idx,
                       h->djet_state, h->njet, cfg.nvar, g.ncell);
  }
  time_end(h, 2);
  HCHECK(h, hipGetLastError());
  if (full) h->ph_valid = false;
  return 0;
}

int pion_gpu_calc_dt_device(void *handle, void **dptr)
{
  Handle *h = use(handle);
  DtArgs a;
  a.g = h->g;
  a.P = h->dP;
  a.Ph = h->ph_valid ? h->dPh : h->dP;
  a.flags = h->dflags;
  a.result = h->ddt;
  a.errword = h->derr;
  a.eqntype = h->cfg.eqntype;
  a.nvar = h->cfg.nvar;
  a.gamma = h->cfg.gamma;
  a.cfl = h->cfg.cfl;
  a.do_mp = mp_dt_limited(h->cfg) ? 1 : 0;
  a.cool = h->cool;
  if (a.do_mp && !h->have_tables) {
    h->err = "cooling tables not set";
    return PION_GPU_EINVAL;
  }
  if (!h->dt_cached) {
    // (after a full step through k_stage_rows2 the minima of the new state are already in ddt)
    HCHECK(h, hipMemcpyAsync(h->ddt, h->ddt_init, 2 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    time_begin(h, 3);
    const int rc = h->cfg.strict_fp ? fp_strict::launch_dt(a, h->stream) : fp_fast::launch_dt(a, h->stream);
    time_end(h, 3);
    if (rc != 0) {
      h->err = "dt kernel launch failed";
      return PION_GPU_EDEVICE;
    }
    h->dt_cached = true;
  }
  if (dptr) *dptr = h->ddt;
  return 0;
}

int pion_gpu_dt_request(void *handle)
{
  Handle *h = use(handle);
  if (!h->hdt) HCHECK(h, hipHostMalloc((void **)&h->hdt, 4 * sizeof(double), hipHostMallocDefault));
  if (!h->ev_dt) HCHECK(h, hipEventCreateWithFlags(&h->ev_dt, hipEventDisableTiming));
  // {min t_dyn, min t_mp} and the device error word, one event for both
  HCHECK(h, hipMemcpyAsync(h->hdt, h->ddt, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHECK(h, hipMemcpyAsync(h->hdt + 2, h->derr, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HCHECK(h, hipEventRecord(h->ev_dt, h->stream));
  h->dt_requested = true;
  return 0;
}

int pion_gpu_dt_wait(void *handle, double *t_dyn, double *t_mp)
{
  Handle *h = use(handle);
  if (!h->dt_requested) {
    h->err = "pion_gpu_dt_wait without pion_gpu_dt_request";
    return PION_GPU_EINVAL;
  }
  HCHECK(h, hipEventSynchronize(h->ev_dt));
  h->dt_requested = false;
  *t_dyn = h->hdt[0];
  *t_mp = h->hdt[1];
  int e;
  memcpy(&e, h->hdt + 2, sizeof e);
  if (e) return check_errword(h);   // (re-reads and clears the word, builds the message)
  return 0;
}

int pion_gpu_read_dt(void *handle, double *t_dyn, double *t_mp)
{
  if (int rc = pion_gpu_dt_request(handle)) return rc;
  return pion_gpu_dt_wait(handle, t_dyn, t_mp);
}

int pion_gpu_calc_dt(void *handle, double *t_dyn, double *t_mp)
{
  if (int rc = pion_gpu_calc_dt_device(handle, nullptr)) return rc;
  return pion_gpu_read_dt(handle, t_dyn, t_mp);
}

void *pion_gpu_get_stream(void *handle, int which)
{
  Handle *h = use(handle);
  return (void *)(which == 0 ? h->stream : (h->comm_stream ? h->comm_stream : h->stream));
}

int pion_gpu_set_glm_speeds(void *handle, double dt, double dx, double cr)
{
  Handle *h = use(handle);
  h->glm_chyp = h->cfg.cfl * dx / dt;  // GLMsetPsiSpeed(FV_cfl*delx/delt, cr)
  h->glm_cr = cr;
  return 0;
}

// One stage, or a part of one (PION_STAGE_WHOLE / _INTERIOR / _ZBOUNDARY).  The split lets the z-halo
// exchange of a slab run under the interior: the interior part reads no z ghost plane, the z-boundary
// part (the nbc on-grid planes next to each z face) waits for the unpacked halo.
static bool stage_can_split(const Handle *h)
{
  return h->use_march != 0 && !h->deta && h->g.ndim == 3
         && h->g.ng[2] > 2 * h->g.nbc[2] && !(h->cfg.tm_ooa == 1 && h->cfg.sp_ooa == 1);
}

// min of the cooling time over the state the full step has just written (P), into ddt[1]: k_dt_mp
static int launch_cooling_time(Handle *h, hipStream_t s)
{
  const pion_gpu_config &cfg = h->cfg;
  DtArgs d;
  d.g = h->g;
  d.P = h->dP;
  d.Ph = h->dP;
  d.flags = h->dflags;
  d.result = h->ddt;
  d.errword = h->derr;
  d.eqntype = cfg.eqntype;
  d.nvar = cfg.nvar;
  d.gamma = cfg.gamma;
  d.cfl = cfg.cfl;
  d.do_mp = 1;
  d.cool = h->cool;
  time_begin(h, 3);
  const int rc = cfg.strict_fp ? fp_strict::launch_dt_mp(d, s) : fp_fast::launch_dt_mp(d, s);
  time_end(h, 3);
  if (rc != 0) {
    h->err = "cooling-time kernel launch failed";
    return PION_GPU_EDEVICE;
  }
  return 0;
}

// planes [kz0,kz1) and, if kz3 > kz2, also [kz2,kz3) (the two z-boundary strips go out as ONE launch)
static int stage_launch(Handle *h, double dt_stage, int space_ooa, int is_full_step, int kz0, int kz1,
                        bool first, bool last, int kz2 = 0, int kz3 = 0, hipStream_t ls = 0, bool use_ls = false)
{
  if (!use_ls) ls = h->stream;   // launch stream: the compute stream unless the caller runs this part beside it
  const pion_gpu_config &cfg = h->cfg;
  if (cfg.cooling != 0 && !h->have_tables) {
    h->err = "cooling tables not set";
    return PION_GPU_EINVAL;
  }
  // the stencil state: Ph.  At the start of a step Ph == P in every cell.
  const double *S = h->ph_valid ? h->dPh : h->dP;
  int rc = 0;
  // preprocess_data (solver_eqn_base.cpp:353-415)
  if (h->dhll || h->deta) {
    PrepassArgs p;
    p.g = h->g;
    p.S = S;
    p.hllflag = h->dhll;
    p.divv = nullptr;
    p.gradp = nullptr;
    p.eta = h->deta;
    p.eqntype = cfg.eqntype;
    p.nvar = cfg.nvar;
    p.space_ooa = space_ooa;
    p.gamma = cfg.gamma;
    // cells whose flag this part is the first to need: the faces of on-grid planes [kz0,kz1) touch
    // planes kz0-1 .. kz1; a whole stage covers every cell incl. ghosts like the reference's loop
    p.c0 = 0;
    p.c1 = h->g.ncell;
    p.c2 = p.c3 = 0;
    if (!(first && last)) {
      const int nb = h->g.nbc[2], nz = h->g.ng[2];
      int lo = kz0 - 1, hi = kz1 + 1;        // on-grid plane numbers [lo,hi): the interior part
      if (kz0 == 0) {                        // lower z-boundary part: the interior part did nb-1 ..
        lo = -1;
        hi = nb - 1;
      }
      else if (kz1 == nz) {                  // upper z-boundary part: the interior part did .. nz-nb
        lo = nz - nb + 1;
        hi = nz + 1;
      }
      p.c0 = (long)(lo + nb) * h->g.sz;
      p.c1 = (long)(hi + nb) * h->g.sz;
    }
    if (kz3 > kz2) {
      // second strip (the upper z boundary): its own plane range of flags, same launch
      const int nb = h->g.nbc[2], nz = h->g.ng[2];
      p.c2 = (long)(nz - nb + 1 + nb) * h->g.sz;
      p.c3 = (long)(nz + 1 + nb) * h->g.sz;
    }
    time_begin(h, 1);
    rc = cfg.strict_fp ? fp_strict::launch_prepass(p, ls) : fp_fast::launch_prepass(p, ls);
    time_end(h, 1);
    if (rc != 0) {
      h->err = "prepass launch failed";
      return PION_GPU_EDEVICE;
    }
  }
  StageArgs a;
  a.g = h->g;
  a.S = S;
  a.Pc = h->dP;
  // first half step: P -> Ph.  Full step: in place on P (each thread reads P only at its own cell).
  a.out = is_full_step ? h->dP : h->dPh;
  if (is_full_step && !h->ph_valid && cfg.sp_ooa == 2 && space_ooa == 2) {
    // a full second-order stage straight from P would read neighbours that are being overwritten
    h->err = "full-step stage requires a preceding half-step stage";
    return PION_GPU_EINVAL;
  }
  if (is_full_step && S == h->dP) {
    // first-order scheme (OA1/OA1): stencil and destination coincide -> go through Ph
    a.out = h->dPh;
  }
  a.flags = h->dflags;
  a.hllflag = h->dhll;
  a.eta = h->deta;
  a.errword = h->derr;
  a.fc = make_fluxctx(h, dt_stage);
  a.eqntype = cfg.eqntype;
  a.ntracer = cfg.ntracer;
  a.solver = cfg.solver;
  a.space_ooa = space_ooa;
  a.cooling = cfg.cooling;
  a.dt = dt_stage;
  a.glm_damp = exp(-dt_stage * h->glm_chyp * h->glm_cr);
  a.max_temp = cfg.max_temp;
  a.cool = h->cool;
  a.use_march = h->use_march;
  a.rows = h->rows;
  a.zchunk = h->zchunk;
  a.kz0 = kz0;
  a.kz1 = kz1;
  a.kz2 = kz2;
  a.kz3 = (kz3 > kz2) ? kz3 : kz2;
  a.zslope_lds = h->zslope_lds;
  a.plain_cells = (h->nwind == 0) ? 1 : 0;
  a.dE = nullptr;
  // periodic x: k_stage_rows2 writes the x ghost images of its rows (the boundary launch then skips them)
  a.xwrap = (a.use_march != 0 && h->fuse_bc && cfg.bc_type[0] == PION_BC_PERIODIC
             && cfg.bc_type[1] == PION_BC_PERIODIC && h->g.ng[0] >= 2 * h->g.nbc[0]) ? 1 : 0;
  a.rows_auto = 0;
  a.ncu = h->ncu;
  if (a.use_march != 0 && h->g.ndim == 2) {
    // 2-D: rows per wavefront marched along y (nothing in LDS): 2 + 1/R solves per cell against the number of
    // wavefronts (measured, 4096 x 1260 Euler Roe-CV / 4096 x 6144 GLM-MHD HLLD, Mcell-updates/s: R = 4 7694 / 5940,
    // 8 11360 / 7711, 16 12686 / 8034, 32 13076 / 7371; the cell-per-thread kernel 4680 / 2176); fewer rows on
    // small grids so that every slot still gets a wavefront (PION_ROWS overrides)
    int r2 = 16;
    {
      const long ntx = (h->g.ng[0] + 61) / 62;
      while (r2 > 2 && ntx * ((h->g.ng[1] + r2 - 1) / r2) < 8L * (h->ncu > 0 ? h->ncu : 256)) r2 /= 2;
    }
    a.rows = (h->rows > 0) ? h->rows : r2;
    if (a.rows > 64) a.rows = 64;
    // (without PION_ROWS the launcher refines the choice for the instance it launches: its occupancy decides how
    // many wavefronts a "round" holds, stage_rows2.h rows2_pick_rows_2d)
    a.rows_auto = (h->rows > 0) ? 0 : 1;
  }
  else if (a.use_march != 0)
    a.rows = cfg.strict_fp ? fp_strict::stage_rows2_rows(cfg.eqntype, cfg.ntracer, a.zslope_lds && space_ooa == 2, space_ooa == 2 ? h->rows : h->rows1)
                           : fp_fast::stage_rows2_rows(cfg.eqntype, cfg.ntracer, a.zslope_lds && space_ooa == 2, space_ooa == 2 ? h->rows : h->rows1);
  if (a.zchunk <= 0) {
    // Planes per wavefront.  Every wavefront takes (zchunk + 1 priming plane) plane visits and a CU holds 8
    // wavefronts at a time; pick the chunk that minimises the launch cost model below.
    int rows = a.rows;
    if (rows < 1) rows = 1;
    const int nyg = (h->g.ng[1] + rows - 1) / rows;
    const int ntx_full = h->g.ng[0] / 62, rem = h->g.ng[0] - ntx_full * 62;
    const int spw = (rem > 0) ? 64 / (rem + 2) : 0;
    const long per_chunk = (long)ntx_full * nyg + ((rem > 0) ? (nyg + spw - 1) / spw : 0);
    const long slots = 8L * (h->ncu > 0 ? h->ncu : 256);   // two workgroups of four wavefronts per CU
    const int np = kz1 - kz0;
    // cost in plane visits: wavefronts are dispatched as slots free up, so a launch takes about
    // (all wave-visits) / slots plus a tail of half a wavefront's length; short chunks balance better, long
    // chunks prime less (measured at 512^3: 16 and 32 planes 27.3 ms/step, 47: 28.4, 64: 28.0, 128: 31.7)
    double best_cost = -1.0;
    a.zchunk = 8;
    for (int zc = 8; zc <= 128; zc++) {
      const long nzc = (np + zc - 1) / zc;
      const int longest = (zc < np ? zc : np) + 1;
      const int last = np - (int)(nzc - 1) * zc;           // planes of the last chunk
      if (nzc > 1 && 4 * last < 3 * zc) continue;          // a short last chunk unbalances the tail (22, 26: measured)
      double cost = (double)per_chunk * (double)(np + nzc) / (double)slots + 0.5 * longest;
      if (np % zc != 0) cost *= 1.005;                     // equal chunks first (512^3: 32 planes 25.5, 27 planes 25.8 ms/step)
      if (best_cost < 0 || cost < best_cost) {
        best_cost = cost;
        a.zchunk = zc;
      }
    }
  }
  // Uneven chunks (default; PION_UNEVEN_CHUNKS=0: equal chunks of a.zchunk planes): chunks of the model's length
  // while more than two of them remain, then halving down to 4 planes.  Wavefronts are dispatched in chunk order, so
  // the last ones to start are the shortest and the launch ends with (nearly) all slots busy; a priming plane costs
  // about a third of a plane visit (its z task only).  512 planes: 14 x 32, 32, 16, 8, 4, 4; a 64-plane slab:
  // 32, 16, 8, 4, 4 (equal chunks: 3.25 ms/step for 512 x 512 x 64, 88 % of the per-cell rate of 512^3).
  a.nzb = 0;
  a.zcmax = 32;
  if (h->zchunk > 0) a.zcmax = h->zchunk;
  else if (a.use_march != 0) {
    // longest chunk: at least ~4 wavefronts per slot over the launch (Euler instances run three workgroups per CU,
    // the MHD ones two), between 8 and 32 planes (256^3 Euler: 11 planes; even chunks of the model 3.05 ms/step,
    // uneven ones from 32 down 3.28)
    int rows = a.rows < 1 ? 1 : a.rows;
    const int nyg = (h->g.ng[1] + rows - 1) / rows;
    const int ntx_full = h->g.ng[0] / 62, rem = h->g.ng[0] - ntx_full * 62;
    const int spw = (rem > 0) ? 64 / (rem + 2) : 0;
    const long per_chunk = (long)ntx_full * nyg + ((rem > 0) ? (nyg + spw - 1) / spw : 0);
    const long slots = ((cfg.eqntype == PION_EQEUL) ? 12L : 8L) * (h->ncu > 0 ? h->ncu : 256);
    long c = (long)(kz1 - kz0) * per_chunk / (4 * slots);
    a.zcmax = (int)(c < 8 ? 8 : (c > 32 ? 32 : c));
  }
  if (a.use_march != 0 && h->uneven_chunks && kz1 - kz0 >= 16) {
    int k0, k1;
    a.nzb = zchunk_bounds(kz1 - kz0, a.zcmax, 0, &k0, &k1);
  }
  if (cfg.cooling != 0 && a.use_march != 0) {
    // calc_noRT_microphysics_dU as its own launch (thread per cell, full occupancy): dE per cell
    a.dE = h->ddE;
    time_begin(h, 1);
    rc = cfg.strict_fp ? fp_strict::launch_cooling_dE(a, ls) : fp_fast::launch_cooling_dE(a, ls);
    time_end(h, 1);
    if (rc != 0) {
      h->err = "cooling kernel launch failed";
      return PION_GPU_EDEVICE;
    }
  }
  // fused time-step reduction: the full stage leaves min(t_dyn), min(t_mp) of the new state in ddt
  // (second-order stages only: the first-order instances of k_stage_rows2 carry no reduction code)
  const bool fuse_dt = h->fuse_dt && is_full_step && space_ooa == 2 && a.use_march != 0
                       && ((h->g.ndim == 3 && h->g.nbc[2] >= 2) || h->g.ndim == 2) && a.out == h->dP;
  a.dtres = nullptr;
  a.cfl = cfg.cfl;
  // the cooling time: not in the stage kernel's fused reduction but in its own launch behind the last part of the
  // stage (k_dt_mp, rate tables in LDS; PION_SPLIT_DT_MP=0: fused, A/B)
  const bool split_mp = fuse_dt && mp_dt_limited(cfg) && h->split_dt_mp;
  a.dt_mp = (mp_dt_limited(cfg) && !split_mp) ? 1 : 0;
  h->dt_cached = false;
  if (fuse_dt) {
    if (first)
      HCHECK(h, hipMemcpyAsync(h->ddt, h->ddt_init, 2 * sizeof(double), hipMemcpyDeviceToDevice, ls));
    a.dtres = h->ddt;
  }
  if (first && !last && h->concurrent_strips && !h->timing && h->comm_stream && h->comm_stream != h->stream) {
    // interior part of a split stage: everything the z-boundary strips depend on besides the halo (the
    // previous stage, its boundary update, this part's flags, the reset of the dt minima) is on the stream up
    // to here -- the strips may run beside the interior kernel from this point on (pion_gpu_stage_part)
    if (!h->ev_pre) HCHECK(h, hipEventCreateWithFlags(&h->ev_pre, hipEventDisableTiming));
    HCHECK(h, hipEventRecord(h->ev_pre, ls));
    h->ev_pre_valid = true;
  }
  time_begin(h, 0);
  rc = cfg.strict_fp ? fp_strict::launch_stage(a, ls) : fp_fast::launch_stage(a, ls);
  time_end(h, 0);
  if (rc != 0) {
    h->err = "stage kernel launch failed (unsupported eqn/solver/tracer combination?)";
    return PION_GPU_EDEVICE;
  }
  if (!last) return 0;
  if (split_mp) {
    // (a part launched on the side stream runs beside the interior part: the launch then follows where the compute
    // stream has joined both, pion_gpu_stage_part)
    if (use_ls) h->dt_mp_pending = true;
    else if (int rc2 = launch_cooling_time(h, ls)) return rc2;
  }
  if (is_full_step && a.out == h->dPh) {
    // OA1/OA1: copy the result back to P ("P = Ph", time_integrator.cpp:938-939)
    const size_t nb = sizeof(double) * (size_t)cfg.nvar * h->g.ncell;
    HCHECK(h, hipMemcpyAsync(h->dP, h->dPh, nb, hipMemcpyDeviceToDevice, ls));
  }
  h->ph_valid = !is_full_step;
  h->dt_cached = fuse_dt;
  h->xghost_fresh = a.xwrap ? ((is_full_step && a.out == h->dPh) ? h->dP : a.out) : nullptr;
  return 0;
}

int pion_gpu_stage_part(void *handle, double dt_stage, int space_ooa, int is_full_step, int part)
{
  Handle *h = use(handle);
  const int nz = h->g.ng[2], nb = h->g.nbc[2];
  if (part == PION_STAGE_WHOLE) {
    if (int rc = order_after_unpack(h)) return rc;
    return stage_launch(h, dt_stage, space_ooa, is_full_step, 0, nz, true, true);
  }
  const bool split = stage_can_split(h);
  if (part == PION_STAGE_INTERIOR) {
    if (!split) return 0;  // everything happens in the z-boundary call
    return stage_launch(h, dt_stage, space_ooa, is_full_step, nb, nz - nb, true, false);
  }
  if (part != PION_STAGE_ZBOUNDARY) return PION_GPU_EINVAL;
  if (split && h->ev_pre_valid && h->ev_unpacked_valid) {
    // Two-stream mode: the strips (4 of the slab's planes: a launch of ~2100 short wavefronts on 2048 slots)
    // go to a third stream that waits for the halo and for the point of the compute stream just before the
    // interior kernel, so that they fill the slots the interior launch leaves idle in its last round instead
    // of running after it; the compute stream continues behind both.
    h->ev_pre_valid = false;
    if (!h->bstream) HCHECK(h, hipStreamCreateWithFlags(&h->bstream, hipStreamNonBlocking));
    if (!h->ev_bdone) HCHECK(h, hipEventCreateWithFlags(&h->ev_bdone, hipEventDisableTiming));
    HCHECK(h, hipStreamWaitEvent(h->bstream, h->ev_pre, 0));
    HCHECK(h, hipStreamWaitEvent(h->bstream, h->ev_unpacked, 0));
    const int rc = stage_launch(h, dt_stage, space_ooa, is_full_step, 0, nb, false, true, nz - nb, nz, h->bstream, true);
    if (rc) return rc;
    HCHECK(h, hipEventRecord(h->ev_bdone, h->bstream));
    HCHECK(h, hipStreamWaitEvent(h->stream, h->ev_bdone, 0));
    if (h->dt_mp_pending) {
      h->dt_mp_pending = false;
      if (int rc2 = launch_cooling_time(h, h->stream)) return rc2;
    }
    return order_after_unpack(h);
  }
  h->ev_pre_valid = false;
  // the z ghost planes must have arrived: order the compute stream after the last unpack
  if (int rc = order_after_unpack(h)) return rc;
  if (!split) return stage_launch(h, dt_stage, space_ooa, is_full_step, 0, nz, true, true);
  return stage_launch(h, dt_stage, space_ooa, is_full_step, 0, nb, false, true, nz - nb, nz);
}

int pion_gpu_stage(void *handle, double dt_stage, int space_ooa, int is_full_step)
{
  return pion_gpu_stage_part(handle, dt_stage, space_ooa, is_full_step, PION_STAGE_WHOLE);
}

int pion_gpu_advance_time(void *handle, double dt, double simtime)
{
  Handle *h = use(handle);
  int rc;
  if (h->cfg.tm_ooa == 1 && h->cfg.sp_ooa == 1) {
    if ((rc = pion_gpu_stage(handle, dt, 1, 1))) return rc;
    return pion_gpu_update_bcs(handle, simtime, 1, 1, 0);
  }
  if (h->cfg.tm_ooa == 2 && h->cfg.sp_ooa == 2) {
    if ((rc = pion_gpu_stage(handle, 0.5 * dt, 1, 0))) return rc;
    if ((rc = pion_gpu_update_bcs(handle, simtime, 1, 2, 0))) return rc;
    if ((rc = pion_gpu_stage(handle, dt, 2, 1))) return rc;
    return pion_gpu_update_bcs(handle, simtime, 2, 2, 0);
  }
  h->err = "Bad OOA requests; choose (1,1) or (2,2)";
  return PION_GPU_EINVAL;
}

long pion_gpu_halo_count(void *handle)
{
  Handle *h = use(handle);
  return (long)h->cfg.nvar * h->g.nbc[2] * h->g.nga[0] * h->g.nga[1];
}
static int halo_go(Handle *h, int which, int face, void *dbuf, int pack)
{
  if (h->cfg.ndim != 3 || (face != 4 && face != 5)) return PION_GPU_EINVAL;
  double *A = (which == 0) ? h->dP : h->dPh;
  const long n = pion_gpu_halo_count(h);
  hipStream_t cs = h->comm_stream ? h->comm_stream : h->stream;
  if (cs != h->stream && pack) {
    // the planes to send were written (stage) and their x/y ghosts filled (BCs) on the compute stream
    if (!h->ev_packed_src) HCHECK(h, hipEventCreateWithFlags(&h->ev_packed_src, hipEventDisableTiming));
    HCHECK(h, hipEventRecord(h->ev_packed_src, h->stream));
    HCHECK(h, hipStreamWaitEvent(cs, h->ev_packed_src, 0));
  }
  hipLaunchKernelGGL(k_halo, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cs, A, (double *)dbuf, h->g,
                     h->cfg.nvar, face, pack);
  HCHECK(h, hipGetLastError());
  if (cs != h->stream && !pack) {
    if (!h->ev_unpacked) HCHECK(h, hipEventCreateWithFlags(&h->ev_unpacked, hipEventDisableTiming));
    HCHECK(h, hipEventRecord(h->ev_unpacked, cs));
    h->ev_unpacked_valid = true;
  }
  return 0;
}
int pion_gpu_halo_spans(void *handle, int which, pion_gpu_halo_spans_t *out)
{
  Handle *h = use(handle);
  if (h->cfg.ndim != 3 || !out) return PION_GPU_EINVAL;
  double *A = (which == 0) ? h->dP : h->dPh;
  const long nb = h->g.nbc[2], nz = h->g.ng[2], sz = h->g.sz;
  out->recv_lo = A;                       // ghost planes 0 .. nb-1
  out->send_lo = A + nb * sz;             // first on-grid planes
  out->send_hi = A + nz * sz;             // last on-grid planes (all-cell planes nz .. nz+nb-1)
  out->recv_hi = A + (nz + nb) * sz;      // ghost planes nz+nb ..
  out->count_per_var = nb * sz;
  out->var_stride = h->g.ncell;
  out->nvar = h->cfg.nvar;
  return 0;
}
int pion_gpu_halo_begin(void *handle)
{
  Handle *h = use(handle);
  hipStream_t cs = h->comm_stream ? h->comm_stream : h->stream;
  if (cs != h->stream) {
    // the planes to send were written (stage) and their x/y ghosts filled (BCs) on the compute stream
    if (!h->ev_packed_src) HCHECK(h, hipEventCreateWithFlags(&h->ev_packed_src, hipEventDisableTiming));
    HCHECK(h, hipEventRecord(h->ev_packed_src, h->stream));
    HCHECK(h, hipStreamWaitEvent(cs, h->ev_packed_src, 0));
  }
  return 0;
}
int pion_gpu_halo_end(void *handle)
{
  Handle *h = use(handle);
  hipStream_t cs = h->comm_stream ? h->comm_stream : h->stream;
  if (cs != h->stream) {
    if (!h->ev_unpacked) HCHECK(h, hipEventCreateWithFlags(&h->ev_unpacked, hipEventDisableTiming));
    HCHECK(h, hipEventRecord(h->ev_unpacked, cs));
    h->ev_unpacked_valid = true;
  }
  return 0;
}
int pion_gpu_pack_halo(void *handle, int which, int face, void *dbuf) { return halo_go(use(handle), which, face, dbuf, 1); }
int pion_gpu_unpack_halo(void *handle, int which, int face, void *dbuf) { return halo_go(use(handle), which, face, dbuf, 0); }

int pion_gpu_interface_flux(void *handle, int n, int axis, double dt, const double *Pl, const double *Pr,
                            const double *aux, double *F, double *Pstar)
{
  Handle *h = use(handle);
  const int nv = h->cfg.nvar;
  DevBuf bl, br, ba, bf, bp;
  const size_t nb = sizeof(double) * (size_t)n * nv;
  HCHECK(h, hipMalloc(&bl.p, nb));
  HCHECK(h, hipMalloc(&br.p, nb));
  HCHECK(h, hipMalloc(&bf.p, nb));
  HCHECK(h, hipMalloc(&bp.p, nb));
  HCHECK(h, hipMalloc(&ba.p, sizeof(double) * 4 * n));
  double *dl = bl.p, *dr = br.p, *da = ba.p, *df = bf.p, *dp = bp.p;
  HCHECK(h, hipMemcpy(dl, Pl, nb, hipMemcpyHostToDevice));
  HCHECK(h, hipMemcpy(dr, Pr, nb, hipMemcpyHostToDevice));
  HCHECK(h, hipMemcpy(da, aux, sizeof(double) * 4 * n, hipMemcpyHostToDevice));
  FluxTestArgs a;
  a.n = n;
  a.axis = axis;
  a.eqntype = h->cfg.eqntype;
  a.ntracer = h->cfg.ntracer;
  a.solver = h->cfg.solver;
  a.Pl = dl;
  a.Pr = dr;
  a.aux = da;
  a.F = df;
  a.Pstar = dp;
  a.errword = h->derr;
  a.fc = make_fluxctx(h, dt);
  int rc = h->cfg.strict_fp ? fp_strict::launch_flux_test(a, h->stream) : fp_fast::launch_flux_test(a, h->stream);
  if (rc != 0) {
    h->err = "interface-flux launch failed";
    return PION_GPU_EDEVICE;
  }
  HCHECK(h, hipStreamSynchronize(h->stream));
  HCHECK(h, hipMemcpy(F, df, nb, hipMemcpyDeviceToHost));
  HCHECK(h, hipMemcpy(Pstar, dp, nb, hipMemcpyDeviceToHost));
  // the physics error word is informational here (tests feed extreme states)
  int z = 0;
  hipMemcpy(h->derr, &z, sizeof(int), hipMemcpyHostToDevice);
  return 0;
}

static int cool_go(Handle *h, int n, double dt, const double *Pin, double *Pout, const double *rho, const double *T,
                   double *edot)
{
  if (!h->have_tables) {
    h->err = "cooling tables not set";
    return PION_GPU_EINVAL;
  }
  CoolTestArgs a;
  memset(&a, 0, sizeof a);
  a.n = n;
  a.nvar = h->cfg.nvar;
  a.dt = dt;
  a.gamma = h->cfg.gamma;
  a.errword = h->derr;
  a.cool = h->cool;
  DevBuf b0, b1, b2;
  double *&d0 = b0.p, *&d1 = b1.p, *&d2 = b2.p;
  int rc;
  if (Pin) {
    const size_t nb = sizeof(double) * (size_t)n * a.nvar;
    HCHECK(h, hipMalloc(&d0, nb));
    HCHECK(h, hipMalloc(&d1, nb));
    HCHECK(h, hipMemcpy(d0, Pin, nb, hipMemcpyHostToDevice));
    a.Pin = d0;
    a.Pout = d1;
    if (edot) {  // cooling time of each state (edot = the output array, n doubles)
      HCHECK(h, hipMalloc(&d2, sizeof(double) * (size_t)n));
      a.edot = d2;
      rc = h->cfg.strict_fp ? fp_strict::launch_cool_timescale(a, h->stream) : fp_fast::launch_cool_timescale(a, h->stream);
      HCHECK(h, hipStreamSynchronize(h->stream));
      HCHECK(h, hipMemcpy(edot, d2, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    }
    else {
      rc = h->cfg.strict_fp ? fp_strict::launch_cool_update(a, h->stream) : fp_fast::launch_cool_update(a, h->stream);
      HCHECK(h, hipStreamSynchronize(h->stream));
      HCHECK(h, hipMemcpy(Pout, d1, nb, hipMemcpyDeviceToHost));
    }
  }
  else {
    const size_t nb = sizeof(double) * (size_t)n;
    HCHECK(h, hipMalloc(&d0, nb));
    HCHECK(h, hipMalloc(&d1, nb));
    HCHECK(h, hipMalloc(&d2, nb));
    HCHECK(h, hipMemcpy(d0, rho, nb, hipMemcpyHostToDevice));
    HCHECK(h, hipMemcpy(d1, T, nb, hipMemcpyHostToDevice));
    a.rho = d0;
    a.T = d1;
    a.edot = d2;
    rc = h->cfg.strict_fp ? fp_strict::launch_cool_edot(a, h->stream) : fp_fast::launch_cool_edot(a, h->stream);
    HCHECK(h, hipStreamSynchronize(h->stream));
    HCHECK(h, hipMemcpy(edot, d2, nb, hipMemcpyDeviceToHost));
  }
  if (rc != 0) return PION_GPU_EDEVICE;
  return check_errword(h);
}
int pion_gpu_cooling_update(void *handle, int n, double dt, const double *P_in, double *P_out)
{
  return cool_go(use(handle), n, dt, P_in, P_out, nullptr, nullptr, nullptr);
}
int pion_gpu_cooling_edot(void *handle, int n, const double *rho, const double *T, double *edot)
{
  return cool_go(use(handle), n, 0.0, nullptr, nullptr, rho, T, edot);
}
int pion_gpu_cooling_timescale(void *handle, int n, const double *P_in, double *t_cool)
{
  return cool_go(use(handle), n, 0.0, P_in, nullptr, nullptr, nullptr, t_cool);
}

int pion_gpu_enable_timing(void *handle, int on)
{
  Handle *h = use(handle);
  h->timing = on != 0;
  for (int s = 0; s < 4; s++) {
    for (hipEvent_t e : h->ev[s]) hipEventDestroy(e);
    h->ev[s].clear();
  }
  return 0;
}
int pion_gpu_get_timing(void *handle, double *out, int n)
{
  Handle *h = use(handle);
  HCHECK(h, hipStreamSynchronize(h->stream));
  for (int s = 0; s < 4 && s < n; s++) {
    double tot = 0.0;
    int cnt = 0;
    for (size_t k = 0; k + 1 < h->ev[s].size(); k += 2) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, h->ev[s][k], h->ev[s][k + 1]) == hipSuccess) {
        tot += ms;
        cnt++;
      }
    }
    out[s] = cnt ? tot / cnt : 0.0;
    if (4 + s < n) out[4 + s] = cnt;
  }
  return 0;
}

}  // extern "C"
