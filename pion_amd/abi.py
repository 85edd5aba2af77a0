"""ctypes view of include/pion_gpu.h (the C-ABI of libpion_gpu.so).

Only plain C types cross this boundary.  The constants mirror the reference's
source/constants.h:238-281,321-326 and boundaries/boundaries.h (our own BC
numbering, see the header).
"""
import ctypes as C
import os

PION_MAX_NVAR = 16

# equation types (constants.h:163-170)
EQEUL, EQMHD, EQGLM = 1, 2, 3
# flux solvers (constants.h:238-246)
FLUX_LF, FLUX_RSlinear, FLUX_RSexact, FLUX_RShybrid, FLUX_RSroe = 0, 1, 2, 3, 4
FLUX_RSroe_pv, FLUX_FVS, FLUX_RS_HLLD, FLUX_RS_HLL = 5, 6, 7, 8
# artificial viscosity (constants.h:321-326)
AV_NONE, AV_FKJ98_1D, AV_HCORRECTION, AV_HCORR_FKJ98 = 0, 1, 3, 4
# boundaries
BC_NONE, BC_PERIODIC, BC_OUTFLOW, BC_INFLOW, BC_REFLECTING, BC_FIXED = 0, 1, 2, 3, 4, 5
BC_ONEWAY_OUT, BC_DMACH, BC_DMACH2, BC_STWIND, BC_SLAB, BC_JET, BC_AXISYMMETRIC, BC_JETREFLECT = 6, 7, 8, 9, 10, 11, 12, 13
BC_NAMES = {
    "periodic": BC_PERIODIC, "outflow": BC_OUTFLOW, "inflow": BC_INFLOW,
    "reflecting": BC_REFLECTING, "fixed": BC_FIXED, "one-way-outflow": BC_ONEWAY_OUT,
    "DMR": BC_DMACH, "slab": BC_SLAB, "axisymmetric": BC_AXISYMMETRIC, "jetreflect": BC_JETREFLECT,
}
COOL_NONE, COOL_WSS09_CIE_LINE_HEAT_COOL = 0, 8
# cell flags
CELL_ISGD, CELL_ISBD, CELL_ISDOMAIN, CELL_TIMESTEP, CELL_ISLEAF = 1, 2, 4, 8, 16
# primitive / conserved indices (constants.h:256-281)
RO, PG, VX, VY, VZ, BX, BY, BZ, SI = range(9)
RHO, ERG, MMX, MMY, MMZ, BBX, BBY, BBZ, PSI = range(9)
OA1, OA2 = 1, 2
STAGE_WHOLE, STAGE_INTERIOR, STAGE_ZBOUNDARY = 0, 1, 2

E_OK, E_INVAL, E_DEVICE, E_PHYSICS, E_NOMEM = 0, -1, -2, -3, -4


class PionGpuConfig(C.Structure):
    _fields_ = [
        ("ndim", C.c_int), ("nvar", C.c_int), ("ntracer", C.c_int), ("eqntype", C.c_int),
        ("solver", C.c_int), ("artvisc", C.c_int), ("sp_ooa", C.c_int), ("tm_ooa", C.c_int),
        ("coord_sys", C.c_int), ("nbc", C.c_int),
        ("ng", C.c_int * 3),
        ("xmin", C.c_double * 3),
        ("dx", C.c_double), ("gamma", C.c_double), ("cfl", C.c_double), ("etav", C.c_double),
        ("min_temp", C.c_double), ("max_temp", C.c_double),
        ("refvec", C.c_double * PION_MAX_NVAR),
        ("bc_type", C.c_int * 6),
        ("bc_dmach2", C.c_int), ("cooling", C.c_int), ("mp_timestep_limit", C.c_int),
        ("strict_fp", C.c_int),
    ]


def make_config(ndim, ng, eqntype, solver, nvar=None, ntracer=0, artvisc=AV_FKJ98_1D, etav=0.1,
                gamma=5.0 / 3.0, cfl=0.3, dx=None, xmin=(0.0, 0.0, 0.0), xmax=None, bcs=None,
                refvec=None, ooa=2, nbc=None, min_temp=0.0, max_temp=1e100, cooling=0,
                mp_timestep_limit=0, bc_dmach2=0, strict_fp=1, coord_sys=1):
    """Build a PionGpuConfig the way get_sim_info / setup_fixed_grid would
    (source/ics/get_sim_info.cpp:72-180; Nbc = 2 for second order, setup_fixed_grid.cpp:183-190)."""
    cfg = PionGpuConfig()
    base = {EQEUL: 5, EQMHD: 8, EQGLM: 9}[eqntype]
    cfg.ndim = ndim
    cfg.ntracer = ntracer
    cfg.nvar = nvar if nvar is not None else base + ntracer
    cfg.eqntype = eqntype
    cfg.solver = solver
    cfg.artvisc = artvisc
    cfg.sp_ooa = cfg.tm_ooa = ooa
    cfg.coord_sys = coord_sys   # 1 Cartesian, 2 cylindrical (z,R)
    cfg.nbc = nbc if nbc is not None else (2 if ooa == 2 else 1)
    ng = list(ng) + [1] * (3 - len(ng))
    for a in range(3):
        cfg.ng[a] = ng[a] if a < ndim else 1
        cfg.xmin[a] = xmin[a] if a < len(xmin) else 0.0
    if dx is None:
        if xmax is None:
            raise ValueError("need dx or xmax")
        dx = (xmax[0] - xmin[0]) / ng[0]  # SimPM.dx = Range[XX]/NG[XX]
    cfg.dx = dx
    cfg.gamma = gamma
    cfg.cfl = cfl
    cfg.etav = etav
    cfg.min_temp = min_temp
    cfg.max_temp = max_temp
    rv = list(refvec) if refvec is not None else [1.0] * cfg.nvar
    for v in range(PION_MAX_NVAR):
        cfg.refvec[v] = rv[v] if v < len(rv) else 0.0
    bcs = bcs if bcs is not None else ["periodic"] * (2 * ndim)
    for d in range(6):
        if d < 2 * ndim:
            b = bcs[d]
            cfg.bc_type[d] = BC_NAMES[b] if isinstance(b, str) else int(b)
        else:
            cfg.bc_type[d] = 0
    cfg.bc_dmach2 = bc_dmach2
    cfg.cooling = cooling
    cfg.mp_timestep_limit = mp_timestep_limit
    cfg.strict_fp = strict_fp
    return cfg


def ng_all(cfg):
    return [cfg.ng[a] + (2 * cfg.nbc if a < cfg.ndim else 0) for a in range(3)]


def ncell_all(cfg):
    n = ng_all(cfg)
    return n[0] * n[1] * n[2]


_HERE = os.path.dirname(os.path.abspath(__file__))


def share_torch_hip_runtime():
    """Call before dlopen-ing anything that links libamdhip64 (libpion_gpu.so, libpion_host.so).

    PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  Whichever HIP runtime is
    loaded first serves the whole process: with torch first, our libraries bind to torch's copy and
    everything shares one runtime (streams, device memory and RCCL interoperate); with our library
    first, a later `import torch` brings a second runtime and reports "No HIP GPUs are available".
    torch carries the multi-GPU path (pion_amd.slab), so it goes first whenever it is installed;
    PION_NO_TORCH=1 skips this for torch-free deployments."""
    if os.environ.get("PION_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass


def library_path():
    # PION_GPU_LIB: an alternative build of the same library (compiler-flag A/B runs)
    return os.environ.get("PION_GPU_LIB") or os.path.join(_HERE, "csrc", "libpion_gpu.so")
