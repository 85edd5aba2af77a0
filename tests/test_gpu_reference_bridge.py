"""the reference-side adapter (pion_amd/host/reference_bridge/pion_gpu_bridge.cpp: a class written
against the REFERENCE's own SimParams / GridBaseClass / cell headers, compiled with -I/root/reference/source by
`make -C oracle ref` into oracle/_ref/libpion_ref_bridge.so) drives the grid object of the reference harness --
the reference's cell lists -- through libpion_gpu.so: gather cell::P in NextPt_All order, time steps on the
device, scatter back.  The cells must then hold what the reference's own loops leave in them, bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

import golden_cases as gc
from pion_amd import abi, driver, problems
from cpu_backends import CpuSim, have_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BRIDGE = os.path.join(ROOT, "oracle", "_ref", "libpion_ref_bridge.so")


@pytest.mark.skipif(not (have_ref() and os.path.exists(BRIDGE)), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("case", ["glm_hlld_3d", "hd_roe_3d_bcs", "dmr_2d", "hd_jet_3d", "cool_fvs_3d", "cyl_glm_hlld", "cyl_glm_jet",
                                  "cyl_hd_hcorr_tr"])
def test_bridge_drives_the_reference_grid(case):
    abi.share_torch_hip_runtime()
    if case == "glm_hlld_3d":
        cfg, P = problems.mhd_blast_generic([20, 12, 10], abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=1)
    elif case == "hd_roe_3d_bcs":
        cfg, P = problems.hd_blast_box([16, 14, 12], solver=abi.FLUX_RSroe, ntracer=1, strict_fp=1)
    elif case in ("hd_jet_3d", "cyl_glm_hlld", "cyl_glm_jet", "cyl_hd_hcorr_tr"):
        # internal JETBC boundary: the bridge hands JP.jetradius / JP.jetstate to pion_gpu_set_jet; the cyl_* cases are
        # cylindrical (z,R) grids (axisymmetric boundary on the axis): the reference's cell lists driven through the
        # CYL instance of the rows kernel
        cfg, P = gc.step_case(case)
    elif case == "cool_fvs_3d":
        # EP.cooling = 8 with the reference's own mp_only_cooling as MP: the bridge builds and hands over the tables
        from cpu_backends import install_ref_rate_curves
        install_ref_rate_curves()
        cfg, P = gc.step_case_c(case)
    else:
        cfg, P = problems.double_mach_reflection(52, strict_fp=1)
    nsteps = 4
    with CpuSim(cfg, "ref") as r:
        gc.step_setup(case, r)
        sc = driver.SimControl(r, cfg)
        sc.init(P)
        sc.time_int(nsteps)
        want, twant, dtwant = r.download(0), sc.simtime, sc.last_dt
    lib = C.CDLL(BRIDGE)
    lib.ref_bridge_time_int.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    with CpuSim(cfg, "ref") as r:
        gc.step_setup(case, r)
        r.upload(P)
        t, ldt = C.c_double(), C.c_double()
        n = lib.ref_bridge_time_int(r.h, 0, 1, nsteps, C.byref(t), C.byref(ldt))
        assert n == nsteps, n
        got = r.download(0)
    assert t.value == twant and ldt.value == dtwant
    assert np.array_equal(got, want)
