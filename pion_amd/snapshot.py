"""Raw snapshot / restart files for the device state.

The reference writes Silo / FITS / text through dataIO/ (dataio_base.cpp:60-440 is the header registry);
none of those libraries is needed to checkpoint the path, so the format here is the minimum that makes a
restart bit-identical: a JSON header (the pion_gpu_config fields + simtime, timestep, last_dt, the
reference's SimParams names where one exists) followed by the fp64 SoA state [nvar][nz_all][ny_all][nx_all]
in the boundary layout of include/pion_gpu.h, ghost cells included.  B is stored in code units (no
sqrt(4 pi) rescaling, cf. dataio_silo.cpp:1468-1492), so a round trip changes no bit.

    write(path, cfg, P, simtime, timestep, last_dt)       read(path) -> (cfg, P, meta)
"""
import json
import struct

import numpy as np

from . import abi

MAGIC = b"PIONRAW1"
_SCALARS = ["ndim", "nvar", "ntracer", "eqntype", "solver", "artvisc", "sp_ooa", "tm_ooa", "coord_sys", "nbc",
            "dx", "gamma", "cfl", "etav", "min_temp", "max_temp", "bc_dmach2", "cooling", "mp_timestep_limit",
            "strict_fp"]
_ARRAYS = ["ng", "xmin", "refvec", "bc_type"]


def _cfg_to_dict(cfg):
    d = {k: getattr(cfg, k) for k in _SCALARS}
    for k in _ARRAYS:
        d[k] = list(getattr(cfg, k))
    return d


def _dict_to_cfg(d):
    cfg = abi.PionGpuConfig()
    for k in _SCALARS:
        setattr(cfg, k, d[k])
    for k in _ARRAYS:
        arr = getattr(cfg, k)
        for i, v in enumerate(d[k]):
            arr[i] = v
    return cfg


def write(path, cfg, P, simtime, timestep, last_dt):
    P = np.ascontiguousarray(P, dtype=np.float64)
    nga = abi.ng_all(cfg)
    assert P.shape == (cfg.nvar, nga[2], nga[1], nga[0]), (P.shape, nga)
    header = {"config": _cfg_to_dict(cfg), "simtime": float(simtime), "timestep": int(timestep),
              "last_dt": float(last_dt), "shape": list(P.shape), "dtype": "<f8",
              "layout": "[nvar][nz_all][ny_all][nx_all], ghosts included, code units"}
    hb = json.dumps(header).encode()
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<q", len(hb)))
        f.write(hb)
        f.write(P.astype("<f8", copy=False).tobytes())


def read(path):
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError("%s is not a PIONRAW1 snapshot" % path)
        (n,) = struct.unpack("<q", f.read(8))
        header = json.loads(f.read(n).decode())
        P = np.frombuffer(f.read(), dtype="<f8").reshape(header["shape"]).copy()
    return _dict_to_cfg(header["config"]), P, header
