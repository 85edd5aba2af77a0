"""Host-side time loop, shaped like the reference's sim_control / calc_timestep /
time_integrator (source/sim_control/sim_control.cpp:202-281, calc_timestep.cpp:68-262,
time_integrator.cpp:72-142), driving a stage-granular backend.

The backend is anything with the pion_gpu call shapes (pion_amd.lib.GpuSim on the
product path).  With `comm` set (a pion_amd.slab.SlabComm) the grid is a z-slab of
a larger domain: the ghost planes of the two z faces are exchanged with the
neighbouring ranks after every stage and the time step is min-reduced over ranks,
replacing source/decomposition/MCMD_control.cpp:231-309, comms/comm_mpi.cpp:287-636
and sim_control_MPI.cpp:482-583 for this path.
"""
from . import abi


class SimControl:
    def __init__(self, sim, cfg, comm=None, finishtime=1e300, min_timestep=0.0):
        self.sim = sim
        self.cfg = cfg
        self.comm = comm
        self.simtime = 0.0
        self.timestep = 0
        self.last_dt = 1e100      # SimParams::last_dt is set large before the first step
        self.dt = 0.0
        self.finishtime = finishtime
        self.min_timestep = min_timestep
        self.first_step_dt_limit = None  # wind / jet limit of calc_dynamics_dt (calc_timestep.cpp:313-323)

    # sim_init::Init (sim_init.cpp:219-267): read data, Ph=P, assign + update boundaries
    def init(self, P, simtime=0.0):
        self.simtime = simtime
        self.sim.upload(P)
        self.update_bcs(self.cfg.tm_ooa, self.cfg.tm_ooa, assign=1)

    def update_bcs(self, cstep, maxstep, assign=0):
        """TimeUpdateInternalBCs + TimeUpdateExternalBCs (+ slab halo exchange, which plays the
        role of BC_update_BCMPI for the z faces, MCMD_boundaries.cpp:122-237).  The exchange is only
        started here; _stage() completes it between the interior and the z-boundary part."""
        self.sim.update_bcs(self.simtime, cstep, maxstep, assign)
        if self.comm is not None:
            self.comm.start(self.sim, which=0 if cstep == maxstep else 1)

    def _stage(self, dt, space_ooa, is_full):
        if self.comm is None:
            self.sim.stage(dt, space_ooa, is_full)
            return
        self.sim.stage_part(dt, space_ooa, is_full, abi.STAGE_INTERIOR)
        self.comm.finish(self.sim)
        self.sim.stage_part(dt, space_ooa, is_full, abi.STAGE_ZBOUNDARY)

    def finish_halo(self):
        """Complete a halo exchange still in flight (before the state is read back)."""
        if self.comm is not None:
            self.comm.finish(self.sim)

    # calc_timestep::calculate_timestep (calc_timestep.cpp:68-153)
    def calculate_timestep(self):
        t_dyn, t_mp = self.sim.calc_dt()
        if self.timestep == 0 and self.first_step_dt_limit is not None:
            t_dyn = min(t_dyn, self.first_step_dt_limit)
        if self.comm is not None:
            t_dyn, t_mp = self.comm.allreduce_min(t_dyn, t_mp)  # sim_control_MPI.cpp:503-504
        dt = min(t_dyn, t_mp)
        if self.cfg.eqntype == abi.EQGLM:
            # Set_GLM_Speeds(td, dx, 0.25/dx) with td = t_dyn (calc_timestep.cpp:119-131)
            self.sim.set_glm_speeds(t_dyn, self.cfg.dx, 0.25 / self.cfg.dx)
        # timestep_checking_and_limiting (calc_timestep.cpp:219-262)
        if dt < self.min_timestep:
            raise RuntimeError("Timestep too short! dt=%g min-step=%g" % (dt, self.min_timestep))
        dt = min(dt, 1.3 * self.last_dt)                 # TIMESTEP_LIMITING
        dt = min(dt, self.finishtime - self.simtime)
        if dt <= 0.0:
            raise RuntimeError("Negative timestep!")
        self.dt = dt
        return dt

    # time_integrator::advance_time (time_integrator.cpp:72-142)
    def advance_time(self):
        dt = self.dt
        if self.cfg.tm_ooa == abi.OA1 and self.cfg.sp_ooa == abi.OA1:
            self._stage(dt, abi.OA1, 1)
            self.update_bcs(abi.OA1, abi.OA1)
        elif self.cfg.tm_ooa == abi.OA2 and self.cfg.sp_ooa == abi.OA2:
            self._stage(0.5 * dt, abi.OA1, 0)
            self.update_bcs(abi.OA1, abi.OA2)
            self._stage(dt, abi.OA2, 1)
            self.update_bcs(abi.OA2, abi.OA2)
        else:
            raise RuntimeError("Bad OOA requests; choose (1,1) or (2,2)")
        self.simtime += dt
        self.last_dt = dt
        self.timestep += 1
        return dt

    # sim_control::Time_Int (sim_control.cpp:202-281) without I/O
    def time_int(self, nsteps=None):
        n = 0
        while self.simtime < self.finishtime and (nsteps is None or n < nsteps):
            self.calculate_timestep()
            self.advance_time()
            n += 1
        self.finish_halo()
        return n
