// pion_gpu_bridge.h -- the reference-side adapter: what a PION maintainer adds to a PION build so that
// sim_control drives libpion_gpu.so instead of its own sweep loops.
//
// Written against the REFERENCE's own headers (SimParams, GridBaseClass, cell, boundary_data), i.e. it is
// compiled inside a PION source tree: here by `make -C oracle ref` into oracle/_ref/libpion_ref_bridge.so with
// -I/root/reference/source, and tested on the GPU box by tests/test_gpu_reference_bridge.py, which lets it
// drive the grid object of oracle/ref_harness.cpp (a GridBaseClass with the reference's cell lists) and
// compares with the reference's own loops on the same grid, bit for bit.
//
// It replaces, with the same member names and argument meaning:
//   time_integrator::advance_time / first_order_update / second_order_update   sim_control/time_integrator.cpp:72-250
//   calc_timestep::calculate_timestep (+ timestep_checking_and_limiting)        sim_control/calc_timestep.cpp:68-262
//   sim_control::Time_Int without I/O                                          sim_control/sim_control.cpp:202-281
// (time_integrator.h itself cannot be included in this container: it pulls grid/uniform_grid.h ->
// tools/interpolate.h -> GSL.  In a PION build the class below is a base of, or a member of, a
// `class sim_control_gpu : public sim_control` whose advance_time / calculate_timestep forward to it.)
//
// Boundaries translated: external PERIODIC, OUTFLOW, INFLOW, REFLECTING, FIXED, ONEWAY_OUT, DMACH, AXISYMMETRIC,
// JETREFLECT; internal DMACH2, JETBC (JP.jetradius / JP.jetstate), STWIND with constant winds (the cells' state at
// gather time).  Any other boundary type throws -- nothing is dropped silently.  With EP.cooling the look-up
// tables are built by pion_host_build_cooling_tables (link libpion_host.so) and handed to the device.
//
// The state lives on the device between steps; cell::P / Ph are gathered once (gather_and_upload) and
// scattered back when the host needs them (download_and_scatter: before output, at the end).
#ifndef PION_GPU_BRIDGE_H
#define PION_GPU_BRIDGE_H

#include "tools/reporting.h"   // (must precede the equation headers, eqns_mhd_adiabatic.h:156 uses rep)

#include "boundaries/boundaries.h"
#include "grid/cell_interface.h"
#include "grid/grid_base_class.h"
#include "sim_params.h"

#include "pion_gpu.h"

class pion_gpu_bridge {
 public:
  // builds the pion_gpu_config from SimParams and the grid's boundary list, creates the handle
  pion_gpu_bridge(class SimParams &par, class GridBaseClass *grid, int device, int strict_fp);
  ~pion_gpu_bridge();

  /// cell::P of every cell (ghosts included) in NextPt_All order = id order x fastest
  /// (uniform_grid.cpp:482-636, 820-844) -> [nvar][nz_all][ny_all][nx_all] -> device; then
  /// assign_boundary_data + TimeUpdate*BCs (sim_init.cpp:246-267)
  int gather_and_upload();
  /// device P -> cell::P and cell::Ph of every cell, dU zeroed (the state after a full step)
  int download_and_scatter();

  int calculate_timestep();   ///< sets par.dt from the device reduction, with the reference's limiting
  int advance_time();         ///< one step of the (1,1) or (2,2) scheme; advances par.simtime / timestep / last_dt
  int Time_Int(int nsteps);   ///< calculate_timestep + advance_time, nsteps times or until par.finishtime

  const pion_gpu_config &config() const { return cfg_; }
  std::string last_error() const;

 private:
  class SimParams &par_;
  class GridBaseClass *grid_;
  pion_gpu_config cfg_;
  void *h_;
  long ncell_;
  std::vector<double> soa_;
  std::vector<class cell *> wind_cells_;   // cells of STWIND boundaries; their state is captured at gather time
};

#endif
