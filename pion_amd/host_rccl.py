"""ctypes view of the C++ host layer above the C-ABI (pion_amd/host/libpion_host.so):
pion_host::sim_control_gpu (the Time_Int / calculate_timestep / advance_time mirror) and
pion_host::slab_comm_rccl (z-slab halo exchange + time-step reduction over RCCL, driven from C++:
ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the communication stream, ncclAllReduce(ncclMin) on
the device-resident minima).  Python only hands over configuration, the initial state and -- for N > 1 -- the
128-byte ncclUniqueId it broadcast; every launch, transfer and reduction of the time loop is issued from C++.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_host = None

UNIQUE_ID_BYTES = 128


def load_host_library():
    global _host
    if _host is not None:
        return _host
    path = os.path.join(_HERE, "host", "libpion_host.so")
    if not os.path.exists(path):
        raise ImportError("%s not found: build it with `make -C pion_amd/host`" % path)
    abi.share_torch_hip_runtime()
    # libpion_host.so NEEDs "libpion_gpu.so" (soname): load the library this process is meant to use first, so
    # that an alternative build named by PION_GPU_LIB (A/B runs) also serves the C++ time loop
    C.CDLL(abi.library_path(), mode=C.RTLD_GLOBAL)
    h = C.CDLL(path)
    h.pion_host_sim_create.argtypes = [C.POINTER(abi.PionGpuConfig), C.c_int, C.POINTER(C.c_void_p)]
    h.pion_host_sim_create_backend.argtypes = [C.POINTER(abi.PionGpuConfig), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    h.pion_host_comm_shm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_void_p, C.POINTER(C.c_void_p)]
    h.pion_host_sim_destroy.argtypes = [C.c_void_p]
    h.pion_host_sim_destroy.restype = None
    h.pion_host_sim_handle.argtypes = [C.c_void_p]
    h.pion_host_sim_handle.restype = C.c_void_p
    h.pion_host_sim_init.argtypes = [C.c_void_p, _dp, C.c_double, C.c_double, C.c_double]
    h.pion_host_sim_time_int.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
    h.pion_host_sim_set_time.argtypes = [C.c_void_p, C.c_int, C.c_double]
    h.pion_host_sim_step.argtypes = [C.c_void_p, _dp]
    h.pion_host_sim_download.argtypes = [C.c_void_p, C.c_int, _dp]
    h.pion_host_sim_set_comm.argtypes = [C.c_void_p, C.c_void_p]
    h.pion_host_sim_finish_halo.argtypes = [C.c_void_p]
    h.pion_host_sim_last_error.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    h.pion_host_comm_unique_id.argtypes = [C.c_void_p]
    h.pion_host_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    h.pion_host_comm_destroy.argtypes = [C.c_void_p]
    h.pion_host_comm_destroy.restype = None
    _host = h
    return h


def new_unique_id():
    """ncclGetUniqueId on this rank (rank 0 calls it and broadcasts the 128 bytes)"""
    buf = (C.c_char * UNIQUE_ID_BYTES)()
    if load_host_library().pion_host_comm_unique_id(buf) != 0:
        raise RuntimeError("ncclGetUniqueId failed")
    return bytes(buf)


class HostSim:
    """pion_host::sim_control_gpu, optionally with a slab communicator: pion_host::slab_comm_rccl (unique_id: the
    128-byte ncclUniqueId) or pion_host::slab_comm_shm (shm_name: POSIX shared-memory name common to the ranks,
    "/name" -- the host-staged transport).  backend: address of a pion_backend table (pion_amd/host/pion_backend.h);
    None = libpion_gpu.so, the product's only backend (tests hand in the oracle's)."""

    def __init__(self, cfg, device=0, rank=0, world=1, periodic_z=True, unique_id=None, shm_name=None, backend=None):
        self.lib = load_host_library()
        self.cfg = cfg
        self.s = C.c_void_p()
        self.comm = C.c_void_p()
        if self.lib.pion_host_sim_create_backend(C.byref(cfg), device, backend, C.byref(self.s)) != 0:
            raise RuntimeError("pion_host_sim_create failed: " + self.last_error())
        if shm_name is not None:
            rc = self.lib.pion_host_comm_shm_create(rank, world, 1 if periodic_z else 0, shm_name.encode(), backend,
                                                    C.byref(self.comm))
            if rc != 0:
                self.close()
                raise RuntimeError("pion_host_comm_shm_create failed rc=%d" % rc)
            rc = self.lib.pion_host_sim_set_comm(self.s, self.comm)
            if rc != 0:
                self.close()
                raise RuntimeError("pion_host_sim_set_comm failed rc=%d" % rc)
        elif unique_id is not None:
            idb = (C.c_char * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
            # RCCL prints a version banner on stdout when a communicator is created: send it to stderr
            # (bench.py's stdout carries exactly one JSON line)
            import sys
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                rc = self.lib.pion_host_comm_create(rank, world, 1 if periodic_z else 0, idb, device, C.byref(self.comm))
            finally:
                os.dup2(saved, 1)
                os.close(saved)
            if rc != 0:
                self.close()
                raise RuntimeError("pion_host_comm_create failed")
            rc = self.lib.pion_host_sim_set_comm(self.s, self.comm)
            if rc != 0:
                self.close()
                raise RuntimeError("pion_host_sim_set_comm failed rc=%d" % rc)
        nga = abi.ng_all(cfg)
        self.shape = (cfg.nvar, nga[2], nga[1], nga[0])

    def last_error(self):
        buf = C.create_string_buffer(600)
        self.lib.pion_host_sim_last_error(self.s, buf, 600)
        return buf.value.decode(errors="replace")

    def gpu_handle(self):
        return self.lib.pion_host_sim_handle(self.s)

    def init(self, P, simtime=0.0, finishtime=1e300, first_step_dt_limit=None, timestep=0, last_dt=1e100):
        """sim_init::Init; on a restart pass the snapshot's simtime / timestep / last_dt"""
        self.lib.pion_host_sim_set_time(self.s, int(timestep), float(last_dt))
        Pc = np.ascontiguousarray(P, dtype=np.float64).reshape(-1)
        rc = self.lib.pion_host_sim_init(self.s, Pc.ctypes.data_as(_dp), simtime, finishtime,
                                         -1.0 if first_step_dt_limit is None else first_step_dt_limit)
        if rc != 0:
            raise RuntimeError("pion_host_sim_init rc=%d: %s" % (rc, self.last_error()))

    def step(self):
        dt = C.c_double()
        rc = self.lib.pion_host_sim_step(self.s, C.byref(dt))
        if rc != 0:
            raise RuntimeError("pion_host_sim_step rc=%d: %s" % (rc, self.last_error()))
        return dt.value

    def finish_halo(self):
        rc = self.lib.pion_host_sim_finish_halo(self.s)
        if rc != 0:
            raise RuntimeError("pion_host_sim_finish_halo rc=%d" % rc)

    def time_int(self, nsteps):
        t, ldt = C.c_double(), C.c_double()
        n = self.lib.pion_host_sim_time_int(self.s, nsteps, C.byref(t), C.byref(ldt))
        if n < 0:
            raise RuntimeError("pion_host_sim_time_int: " + self.last_error())
        return n, t.value, ldt.value

    def download(self, which=0):
        out = np.empty(int(np.prod(self.shape)))
        rc = self.lib.pion_host_sim_download(self.s, which, out.ctypes.data_as(_dp))
        if rc != 0:
            raise RuntimeError("pion_host_sim_download rc=%d" % rc)
        return out.reshape(self.shape)

    def close(self):
        # the communicator detaches its stream from the handle: destroy it first
        if self.comm:
            self.lib.pion_host_comm_destroy(self.comm)
            self.comm = C.c_void_p()
        if self.s:
            self.lib.pion_host_sim_destroy(self.s)
            self.s = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
