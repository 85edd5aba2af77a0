"""Host-side cooling tables for mp_only_cooling (EP.cooling = 8): thin wrapper over
pion_amd/host/libpion_host.so (cooling_tables.cpp), which restates
mp_only_cooling::gen_mpoc_lookup_tables (microphysics/mp_only_cooling.cpp:528-579)."""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _load():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "host", "libpion_host.so")
        if not os.path.exists(path):
            raise ImportError("%s not found: run __graft_entry__.build()" % path)
        abi.share_torch_hip_runtime()
        _lib = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        _lib.pion_host_build_cooling_tables.argtypes = [C.c_double, C.c_double, C.c_int, dp, dp, dp]
        for f in ("pion_host_cooling_rate_wss09", "pion_host_hii_rrr", "pion_host_hii_total_cooling"):
            getattr(_lib, f).restype = C.c_double
            getattr(_lib, f).argtypes = [C.c_double]
    return _lib


def build_tables(min_temp, max_temp, nT=200):
    """Returns (T[nT], tabs[5,nT], slopes[5,nT]); rows: rrhp, C_rrh, C_ffhe, C_fbdn, C_cie."""
    lib = _load()
    T = np.zeros(nT)
    tabs = np.zeros(5 * nT)
    slopes = np.zeros(5 * nT)
    dp = C.POINTER(C.c_double)
    rc = lib.pion_host_build_cooling_tables(min_temp, max_temp, nT, T.ctypes.data_as(dp),
                                            tabs.ctypes.data_as(dp), slopes.ctypes.data_as(dp))
    if rc != 0:
        raise ValueError("bad temperature range")
    return T, tabs.reshape(5, nT), slopes.reshape(5, nT)


def cooling_rate_wss09(T):
    return _load().pion_host_cooling_rate_wss09(float(T))


def hii_rrr(T):
    return _load().pion_host_hii_rrr(float(T))


def hii_total_cooling(T):
    return _load().pion_host_hii_total_cooling(float(T))
