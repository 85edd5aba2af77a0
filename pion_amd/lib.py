"""Thin ctypes wrapper around libpion_gpu.so (include/pion_gpu.h).

There is no CPU fallback: if the HIP library is missing or a GPU call fails the
wrapper raises.  Method names mirror the C-ABI, which mirrors the reference's
time_integrator / FV_solver_base entry points (see the header for file:line).
"""
import ctypes as C
import os

import numpy as np

from . import abi

_dp = C.POINTER(C.c_double)
_lib = None


class PionGpuError(RuntimeError):
    def __init__(self, what, rc, msg):
        super().__init__("pion_gpu_%s failed rc=%d: %s" % (what, rc, msg))
        self.rc = rc


def load_library():
    """Load pion_amd/csrc/libpion_gpu.so; loud failure when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = abi.library_path()
    if not os.path.exists(path):
        raise ImportError(
            "%s not found: build it with `make -C pion_amd/csrc` (or __graft_entry__.build()); "
            "pion_amd has no CPU fallback" % path)
    abi.share_torch_hip_runtime()
    lib = C.CDLL(path)
    lib.pion_gpu_create.argtypes = [C.POINTER(abi.PionGpuConfig), C.c_int, C.POINTER(C.c_void_p)]
    lib.pion_gpu_destroy.argtypes = [C.c_void_p]
    lib.pion_gpu_destroy.restype = None
    lib.pion_gpu_last_error.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.pion_gpu_ncell_all.argtypes = [C.c_void_p]
    lib.pion_gpu_ncell_all.restype = C.c_long
    lib.pion_gpu_upload.argtypes = [C.c_void_p, _dp]
    lib.pion_gpu_download.argtypes = [C.c_void_p, C.c_int, _dp]
    lib.pion_gpu_bind_device_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pion_gpu_device_ptr.argtypes = [C.c_void_p, C.c_int]
    lib.pion_gpu_device_ptr.restype = C.c_void_p
    lib.pion_gpu_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.pion_gpu_set_comm_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.pion_gpu_set_jet.argtypes = [C.c_void_p, C.c_int, _dp]
    lib.pion_gpu_stage_part.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int]
    lib.pion_gpu_synchronize.argtypes = [C.c_void_p]
    lib.pion_gpu_set_wind_cells.argtypes = [C.c_void_p, C.c_long, C.POINTER(C.c_long), _dp]
    lib.pion_gpu_set_cooling_tables.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp]
    lib.pion_gpu_update_bcs.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int]
    lib.pion_gpu_calc_dt.argtypes = [C.c_void_p, _dp, _dp]
    lib.pion_gpu_set_glm_speeds.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
    lib.pion_gpu_stage.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int]
    lib.pion_gpu_advance_time.argtypes = [C.c_void_p, C.c_double, C.c_double]
    lib.pion_gpu_halo_count.argtypes = [C.c_void_p]
    lib.pion_gpu_halo_count.restype = C.c_long
    lib.pion_gpu_pack_halo.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.pion_gpu_unpack_halo.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.pion_gpu_interface_flux.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, _dp, _dp, _dp, _dp, _dp]
    lib.pion_gpu_cooling_update.argtypes = [C.c_void_p, C.c_int, C.c_double, _dp, _dp]
    lib.pion_gpu_cooling_edot.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp]
    lib.pion_gpu_cooling_timescale.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
    lib.pion_gpu_enable_timing.argtypes = [C.c_void_p, C.c_int]
    lib.pion_gpu_get_timing.argtypes = [C.c_void_p, _dp, C.c_int]
    _lib = lib
    return lib


# every symbol include/pion_gpu.h declares (checked by the CPU test-suite)
EXPORTED_SYMBOLS = [
    "pion_gpu_create", "pion_gpu_destroy", "pion_gpu_last_error", "pion_gpu_ncell_all",
    "pion_gpu_ng_all", "pion_gpu_upload", "pion_gpu_download", "pion_gpu_bind_device_state",
    "pion_gpu_device_ptr", "pion_gpu_set_stream", "pion_gpu_synchronize", "pion_gpu_set_wind_cells",
    "pion_gpu_set_cooling_tables", "pion_gpu_update_bcs", "pion_gpu_calc_dt",
    "pion_gpu_set_glm_speeds", "pion_gpu_stage", "pion_gpu_advance_time", "pion_gpu_halo_count",
    "pion_gpu_pack_halo", "pion_gpu_unpack_halo", "pion_gpu_interface_flux",
    "pion_gpu_cooling_update", "pion_gpu_cooling_edot", "pion_gpu_cooling_timescale", "pion_gpu_enable_timing",
    "pion_gpu_calc_dt_device", "pion_gpu_read_dt", "pion_gpu_get_stream", "pion_gpu_dt_request", "pion_gpu_dt_wait",
    "pion_gpu_halo_spans", "pion_gpu_halo_begin", "pion_gpu_halo_end",
    "pion_gpu_get_timing", "pion_gpu_stage_part", "pion_gpu_set_comm_stream", "pion_gpu_set_jet",
]


def _p(a):
    return a.ctypes.data_as(_dp)


class GpuSim:
    """One pion_gpu handle (one GPU)."""

    def __init__(self, cfg, device=0, borrowed_handle=None):
        """borrowed_handle: wrap a handle that somebody else owns (pion_host::sim_control_gpu's, see
        pion_amd/host_rccl.py) -- set-up calls and timing only; close() then leaves it alone."""
        self.lib = load_library()
        self.cfg = cfg
        self.owner = borrowed_handle is None
        if borrowed_handle is not None:
            self.h = C.c_void_p(borrowed_handle)
        else:
            self.h = C.c_void_p()
            rc = self.lib.pion_gpu_create(C.byref(cfg), device, C.byref(self.h))
            if rc != 0:
                msg = self._err() if self.h else "invalid configuration"
                self.h = None
                raise PionGpuError("create", rc, msg)
        self.nvar = cfg.nvar
        self.ncell = abi.ncell_all(cfg)
        nga = abi.ng_all(cfg)
        self.shape = (cfg.nvar, nga[2], nga[1], nga[0])

    def _err(self):
        buf = C.create_string_buffer(512)
        self.lib.pion_gpu_last_error(self.h, buf, 512)
        return buf.value.decode()

    def _chk(self, rc, what):
        if rc != 0:
            raise PionGpuError(what, rc, self._err())

    def close(self):
        if getattr(self, "h", None):
            if self.owner:
                self.lib.pion_gpu_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- state
    def upload(self, P):
        P = np.ascontiguousarray(P, dtype=np.float64).reshape(-1)
        assert P.size == self.nvar * self.ncell
        self._chk(self.lib.pion_gpu_upload(self.h, _p(P)), "upload")

    def download(self, which=0):
        out = np.empty(self.nvar * self.ncell)
        self._chk(self.lib.pion_gpu_download(self.h, which, _p(out)), "download")
        return out.reshape(self.shape)

    def bind_device_state(self, dP_ptr, dPh_ptr):
        self._chk(self.lib.pion_gpu_bind_device_state(self.h, C.c_void_p(dP_ptr), C.c_void_p(dPh_ptr)),
                  "bind_device_state")

    def device_ptr(self, which):
        return self.lib.pion_gpu_device_ptr(self.h, which)

    def set_stream(self, stream_ptr):
        self._chk(self.lib.pion_gpu_set_stream(self.h, C.c_void_p(stream_ptr)), "set_stream")

    def set_comm_stream(self, stream_ptr):
        self._chk(self.lib.pion_gpu_set_comm_stream(self.h, C.c_void_p(stream_ptr)), "set_comm_stream")

    def synchronize(self):
        self._chk(self.lib.pion_gpu_synchronize(self.h), "synchronize")

    def set_wind_cells(self, idx, states):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        states = np.ascontiguousarray(states, dtype=np.float64)
        self._chk(self.lib.pion_gpu_set_wind_cells(self.h, idx.size, idx.ctypes.data_as(C.POINTER(C.c_long)),
                                                   _p(states)), "set_wind_cells")

    def set_jet(self, jetradius, jetstate):
        st = np.ascontiguousarray(jetstate, dtype=np.float64)
        self._chk(self.lib.pion_gpu_set_jet(self.h, int(jetradius), _p(st)), "set_jet")

    def set_cooling_tables(self, T, tabs, slopes):
        T = np.ascontiguousarray(T, dtype=np.float64)
        tabs = np.ascontiguousarray(tabs, dtype=np.float64)
        slopes = np.ascontiguousarray(slopes, dtype=np.float64)
        self._chk(self.lib.pion_gpu_set_cooling_tables(self.h, T.size, _p(T), _p(tabs), _p(slopes)),
                  "set_cooling_tables")

    # --- the hot path
    def update_bcs(self, simtime=0.0, cstep=2, maxstep=2, assign=0):
        self._chk(self.lib.pion_gpu_update_bcs(self.h, simtime, cstep, maxstep, assign), "update_bcs")

    def calc_dt(self):
        a, b = C.c_double(), C.c_double()
        self._chk(self.lib.pion_gpu_calc_dt(self.h, C.byref(a), C.byref(b)), "calc_dt")
        return a.value, b.value

    def set_glm_speeds(self, dt, dx, cr):
        self._chk(self.lib.pion_gpu_set_glm_speeds(self.h, dt, dx, cr), "set_glm_speeds")

    def stage(self, dt, space_ooa, is_full):
        self._chk(self.lib.pion_gpu_stage(self.h, dt, space_ooa, is_full), "stage")

    def stage_part(self, dt, space_ooa, is_full, part):
        self._chk(self.lib.pion_gpu_stage_part(self.h, dt, space_ooa, is_full, part), "stage_part")

    def advance_time(self, dt, simtime):
        self._chk(self.lib.pion_gpu_advance_time(self.h, dt, simtime), "advance_time")

    # --- slab halos
    def halo_count(self):
        return self.lib.pion_gpu_halo_count(self.h)

    def pack_halo(self, which, face, dbuf_ptr):
        self._chk(self.lib.pion_gpu_pack_halo(self.h, which, face, C.c_void_p(dbuf_ptr)), "pack_halo")

    def unpack_halo(self, which, face, dbuf_ptr):
        self._chk(self.lib.pion_gpu_unpack_halo(self.h, which, face, C.c_void_p(dbuf_ptr)), "unpack_halo")

    # --- seams
    def interface_flux(self, axis, Pl, Pr, aux=None, dt=1.0):
        Pl = np.ascontiguousarray(Pl, dtype=np.float64)
        Pr = np.ascontiguousarray(Pr, dtype=np.float64)
        n = Pl.shape[0]
        if aux is None:
            aux = np.zeros((n, 4))
        aux = np.ascontiguousarray(aux, dtype=np.float64)
        F = np.zeros((n, self.nvar))
        Ps = np.zeros((n, self.nvar))
        self._chk(self.lib.pion_gpu_interface_flux(self.h, n, axis, dt, _p(Pl), _p(Pr), _p(aux), _p(F), _p(Ps)),
                  "interface_flux")
        return F, Ps

    def cooling_update(self, Pin, dt):
        Pin = np.ascontiguousarray(Pin, dtype=np.float64)
        out = np.zeros_like(Pin)
        self._chk(self.lib.pion_gpu_cooling_update(self.h, Pin.shape[0], dt, _p(Pin), _p(out)), "cooling_update")
        return out

    def cooling_edot(self, rho, T):
        rho = np.ascontiguousarray(rho, dtype=np.float64)
        T = np.ascontiguousarray(T, dtype=np.float64)
        out = np.zeros_like(rho)
        self._chk(self.lib.pion_gpu_cooling_edot(self.h, rho.size, _p(rho), _p(T), _p(out)), "cooling_edot")
        return out

    def cooling_timescale(self, Pin):
        Pin = np.ascontiguousarray(Pin, dtype=np.float64)
        out = np.zeros(Pin.shape[0])
        self._chk(self.lib.pion_gpu_cooling_timescale(self.h, Pin.shape[0], _p(Pin), _p(out)), "cooling_timescale")
        return out

    def enable_timing(self, on=True):
        self._chk(self.lib.pion_gpu_enable_timing(self.h, int(on)), "enable_timing")

    def get_timing(self):
        out = np.zeros(8)
        self._chk(self.lib.pion_gpu_get_timing(self.h, _p(out), 8), "get_timing")
        return {"stage_ms": out[0], "prepass_ms": out[1], "bc_ms": out[2], "dt_ms": out[3],
                "stage_n": int(out[4]), "prepass_n": int(out[5]), "bc_n": int(out[6]), "dt_n": int(out[7])}
