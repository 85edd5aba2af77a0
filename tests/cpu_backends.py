"""ctypes wrappers for the two CPU checkers (TEST INFRASTRUCTURE):

  * oracle/liboracle.so       -- our CPU restatement (prefix orc_)
  * oracle/_ref/libpion_ref.so -- the reference's own solver objects driven by
                                  oracle/ref_harness.cpp (prefix ref_); only
                                  exists where oracle/Makefile `ref` was run.
Both expose the same call shapes as include/pion_gpu.h.
"""
import ctypes as C
import os

import numpy as np

from pion_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libpion_ref.so")

_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


def have_oracle():
    return os.path.exists(ORACLE_SO)


def have_ref():
    return os.path.exists(REF_SO)


_curves_installed = False


def install_ref_rate_curves():
    """Hands the three spline-backed rate curves (WSS09 metals-only CIE cooling, Hummer94 case-B
    recombination rate and total H+ cooling) to the test doubles of oracle/ref_cooling.cpp.  The curves
    are the PRODUCT's (pion_amd/host/cooling_tables.cpp); their values are what stays parity-unpinned.
    From here on a "ref" handle with cfg.cooling != 0 holds the reference's own mp_only_cooling."""
    global _curves_installed
    if _curves_installed:
        return
    from pion_amd import cooling
    host = cooling._load()
    ref = C.CDLL(REF_SO)
    ptr = [C.cast(getattr(host, f), C.c_void_p) for f in
           ("pion_host_cooling_rate_wss09", "pion_host_hii_rrr", "pion_host_hii_total_cooling")]
    ref.ref_cooling_set_rate_curves.restype = None
    ref.ref_cooling_set_rate_curves(*ptr)
    _curves_installed = True


class RefCooling:
    """The reference's mp_only_cooling object alone (oracle/ref_cooling.cpp), EP.cooling = 8."""

    def __init__(self, min_temp, max_temp, gamma, nvar=5, ntracer=0):
        install_ref_rate_curves()
        self.lib = C.CDLL(REF_SO)
        self.h = C.c_void_p()
        self.nvar = nvar
        rc = self.lib.ref_cooling_create(C.c_double(min_temp), C.c_double(max_temp), C.c_double(gamma),
                                         C.c_int(nvar), C.c_int(ntracer), C.byref(self.h))
        if rc != 0:
            raise RuntimeError("ref_cooling_create %d" % rc)

    def close(self):
        if self.h:
            self.lib.ref_cooling_destroy.restype = None
            self.lib.ref_cooling_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def tables(self, nT=200):
        T, tabs, sl = np.zeros(nT), np.zeros((5, nT)), np.zeros((5, nT))
        rc = self.lib.ref_cooling_tables(self.h, C.c_int(nT), _p(T), _p(tabs), _p(sl))
        if rc != 0:
            raise RuntimeError("the reference's table has %d points" % rc)
        return T, tabs, sl

    def limits(self):
        out = np.zeros(3)
        self.lib.ref_cooling_limits.restype = None
        self.lib.ref_cooling_limits(self.h, _p(out))
        return out

    def edot(self, rho, T):
        rho = np.ascontiguousarray(rho, dtype=np.float64)
        T = np.ascontiguousarray(T, dtype=np.float64)
        out = np.zeros_like(rho)
        self.lib.ref_cooling_edot(self.h, C.c_int(rho.size), _p(rho), _p(T), _p(out))
        return out

    def update(self, Pin, dt):
        Pin = np.ascontiguousarray(Pin, dtype=np.float64)
        out, Tf = np.zeros_like(Pin), np.zeros(Pin.shape[0])
        rc = self.lib.ref_cooling_update(self.h, C.c_int(Pin.shape[0]), C.c_double(dt), _p(Pin), _p(out), _p(Tf))
        if rc != 0:
            raise RuntimeError("TimeUpdateMP returned %d" % rc)
        return out, Tf

    def timescale(self, Pin):
        Pin = np.ascontiguousarray(Pin, dtype=np.float64)
        out = np.zeros(Pin.shape[0])
        self.lib.ref_cooling_timescale(self.h, C.c_int(Pin.shape[0]), _p(Pin), _p(out))
        return out


class CpuSim:
    """One handle of the oracle ("orc") or of the reference harness ("ref")."""

    def __init__(self, cfg, kind="orc", borrowed_handle=None):
        self.kind = kind
        self.cfg = cfg
        self.lib = C.CDLL(ORACLE_SO if kind == "orc" else REF_SO)
        self.pre = kind + "_"
        self.h = C.c_void_p()
        self.borrowed = borrowed_handle is not None
        if self.borrowed:
            # an oracle handle owned by someone else (the test backend of the C++ host loop): set-up calls only
            self.h = C.c_void_p(borrowed_handle)
        else:
            rc = self._f("create")(C.byref(cfg), C.byref(self.h))
            if rc != 0:
                raise RuntimeError("create failed %d" % rc)
        self.nvar = cfg.nvar
        self.ncell = abi.ncell_all(cfg)
        nga = abi.ng_all(cfg)
        self.shape = (cfg.nvar, nga[2], nga[1], nga[0])

    def _f(self, name):
        f = getattr(self.lib, self.pre + name)
        f.restype = C.c_int
        return f

    def _chk(self, rc, what):
        if rc != 0:
            msg = ""
            if self.kind == "orc":
                buf = C.create_string_buffer(512)
                self.lib.orc_last_error(self.h, buf, 512)
                msg = buf.value.decode()
            raise RuntimeError("%s%s failed rc=%d %s" % (self.pre, what, rc, msg))

    def close(self):
        if self.h and self.borrowed:
            self.h = None
        if self.h:
            f = getattr(self.lib, self.pre + "destroy")
            f.restype = None
            f(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # --- state
    def upload(self, P):
        P = np.ascontiguousarray(P, dtype=np.float64).reshape(-1)
        assert P.size == self.nvar * self.ncell
        self._chk(self._f("upload")(self.h, _p(P)), "upload")

    def download(self, which=0):
        out = np.empty(self.nvar * self.ncell)
        self._chk(self._f("download")(self.h, C.c_int(which), _p(out)), "download")
        return out.reshape(self.shape)

    def upload_which(self, which, A):
        A = np.ascontiguousarray(A, dtype=np.float64).reshape(-1)
        self._chk(self._f("upload_which")(self.h, C.c_int(which), _p(A)), "upload_which")

    # slab halos on the CPU (oracle only): buffers are numpy arrays passed by address
    def halo_count(self):
        nga = abi.ng_all(self.cfg)
        return self.nvar * self.cfg.nbc * nga[0] * nga[1]

    def _halo_view(self, ptr):
        n = self.halo_count()
        nga = abi.ng_all(self.cfg)
        buf = (C.c_double * n).from_address(ptr)
        return np.frombuffer(buf, dtype=np.float64).reshape(self.nvar, self.cfg.nbc, nga[1], nga[0])

    def pack_halo(self, which, face, ptr):
        A = self.download(which)
        nb, nz = self.cfg.nbc, self.cfg.ng[2]
        v = self._halo_view(ptr)
        v[...] = A[:, nb:2 * nb] if face == 4 else A[:, nz:nz + nb]

    def unpack_halo(self, which, face, ptr):
        A = self.download(which)
        nb, nz = self.cfg.nbc, self.cfg.ng[2]
        v = self._halo_view(ptr)
        if face == 4:
            A[:, 0:nb] = v
        else:
            A[:, nb + nz:nb + nz + nb] = v
        self.upload_which(which, A)
        if which == 0:
            # full step: the reference sets P=Ph in the received ghost cells
            # (MCMD_boundaries.cpp:215-224); the oracle keeps both arrays, so mirror it
            B = self.download(1)
            if face == 4:
                B[:, 0:nb] = v
            else:
                B[:, nb + nz:nb + nz + nb] = v
            self.upload_which(1, B)

    def synchronize(self):
        pass

    def flags(self):
        out = np.empty(self.ncell, dtype=np.uint8)
        self._f("get_flags")(self.h, out.ctypes.data_as(C.POINTER(C.c_ubyte)))
        return out.reshape(self.shape[1:])

    def aux(self, which):
        out = np.empty(self.ncell)
        self._f("get_aux")(self.h, C.c_int(which), _p(out))
        return out.reshape(self.shape[1:])

    def set_wind_cells(self, idx, states):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        states = np.ascontiguousarray(states, dtype=np.float64)
        self._chk(self._f("set_wind_cells")(self.h, C.c_long(idx.size),
                                            idx.ctypes.data_as(C.POINTER(C.c_long)), _p(states)),
                  "set_wind_cells")

    def set_jet(self, jetradius, jetstate):
        st = np.ascontiguousarray(jetstate, dtype=np.float64)
        self._chk(self._f("set_jet")(self.h, C.c_int(int(jetradius)), _p(st)), "set_jet")

    def set_cooling_tables(self, T, tabs, slopes):
        T = np.ascontiguousarray(T, dtype=np.float64)
        tabs = np.ascontiguousarray(tabs, dtype=np.float64)
        slopes = np.ascontiguousarray(slopes, dtype=np.float64)
        self._chk(self._f("set_cooling_tables")(self.h, C.c_int(T.size), _p(T), _p(tabs), _p(slopes)),
                  "set_cooling_tables")

    # --- the path
    def update_bcs(self, simtime=0.0, cstep=2, maxstep=2, assign=0):
        self._chk(self._f("update_bcs")(self.h, C.c_double(simtime), C.c_int(cstep), C.c_int(maxstep),
                                        C.c_int(assign)), "update_bcs")

    def calc_dt(self):
        a, b = C.c_double(), C.c_double()
        self._chk(self._f("calc_dt")(self.h, C.byref(a), C.byref(b)), "calc_dt")
        return a.value, b.value

    def set_glm_speeds(self, dt, dx, cr):
        self._chk(self._f("set_glm_speeds")(self.h, C.c_double(dt), C.c_double(dx), C.c_double(cr)),
                  "set_glm_speeds")

    def stage(self, dt, space_ooa, is_full):
        self._chk(self._f("stage")(self.h, C.c_double(dt), C.c_int(space_ooa), C.c_int(is_full)), "stage")

    def stage_part(self, dt, space_ooa, is_full, part):
        # the oracle has no split: like the configurations pion_gpu_stage_part does not split,
        # everything happens in the z-boundary call (abi.STAGE_ZBOUNDARY), after the halo arrived
        if part != 1:
            self.stage(dt, space_ooa, is_full)

    def setdt(self, dt):
        self._chk(self._f("setdt")(self.h, C.c_double(dt)), "setdt")

    def preprocess(self, csp):
        self._chk(self._f("preprocess")(self.h, C.c_int(csp)), "preprocess")

    def set_dynamics_dU(self, dt, step):
        self._chk(self._f("set_dynamics_dU")(self.h, C.c_double(dt), C.c_int(step)), "set_dynamics_dU")

    def advance_time(self, dt, simtime):
        self._chk(self._f("advance_time")(self.h, C.c_double(dt), C.c_double(simtime)), "advance_time")

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
        self._chk(self._f("interface_flux")(self.h, C.c_int(n), C.c_int(axis), C.c_double(dt), _p(Pl),
                                            _p(Pr), _p(aux), _p(F), _p(Ps)), "interface_flux")
        return F, Ps

    def cell_advance(self, Pin, dU, fv_dt=0.0):
        Pin = np.ascontiguousarray(Pin, dtype=np.float64)
        dU = np.ascontiguousarray(dU, dtype=np.float64)
        out = np.zeros_like(Pin)
        self._chk(self._f("cell_advance")(self.h, C.c_int(Pin.shape[0]), C.c_double(fv_dt), _p(Pin),
                                          _p(dU), _p(out)), "cell_advance")
        return out

    def cell_timestep(self, Pin):
        Pin = np.ascontiguousarray(Pin, dtype=np.float64)
        out = np.zeros(Pin.shape[0])
        self._chk(self._f("cell_timestep")(self.h, C.c_int(Pin.shape[0]), _p(Pin), _p(out)),
                  "cell_timestep")
        return out

    def cooling_update(self, Pin, dt):
        Pin = np.ascontiguousarray(Pin, dtype=np.float64)
        out = np.zeros_like(Pin)
        self._chk(self._f("cooling_update")(self.h, C.c_int(Pin.shape[0]), C.c_double(dt), _p(Pin),
                                            _p(out)), "cooling_update")
        return out

    def cooling_edot(self, rho, T):
        rho = np.ascontiguousarray(rho, dtype=np.float64)
        T = np.ascontiguousarray(T, dtype=np.float64)
        out = np.zeros_like(rho)
        self._chk(self._f("cooling_edot")(self.h, C.c_int(rho.size), _p(rho), _p(T), _p(out)),
                  "cooling_edot")
        return out

    def cooling_timescale(self, Pin):
        Pin = np.ascontiguousarray(Pin, dtype=np.float64)
        out = np.zeros(Pin.shape[0])
        self._chk(self._f("cooling_timescale")(self.h, C.c_int(Pin.shape[0]), _p(Pin), _p(out)),
                  "cooling_timescale")
        return out
