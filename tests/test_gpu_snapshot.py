"""Snapshot / restart on the GPU path: write after two steps, read back into a fresh handle,
continue -- bit-identical to the uninterrupted run, for both builds, also through the C++ host driver."""
import numpy as np
import pytest

from pion_amd import abi, driver, problems, snapshot

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("strict", [1, 0])
@pytest.mark.parametrize("case", ["glm_hlld", "hd_roe_tr", "wind_cooling"])
def test_gpu_snapshot_restart_is_bit_identical(tmp_path, case, strict):
    from pion_amd import cooling, lib
    setup, dtl = None, None
    if case == "glm_hlld":
        cfg, P = problems.mhd_blast_generic([70, 10, 12], abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=strict)
    elif case == "hd_roe_tr":
        cfg, P = problems.hd_blast_box([20, 14, 12], solver=abi.FLUX_RSroe, ntracer=1, strict_fp=strict)
    else:
        cfg, P, (idx, st), dtl = problems.wind3d(16, strict_fp=strict)
        T, tabs, sl = cooling.build_tables(cfg.min_temp, cfg.max_temp)

        def setup(s):
            s.set_cooling_tables(T, tabs, sl)
            s.set_wind_cells(idx, st)
    path = str(tmp_path / "s.pionraw")
    with lib.GpuSim(cfg, 0) as g:
        if setup:
            setup(g)
        sc = driver.SimControl(g, cfg)
        sc.first_step_dt_limit = dtl
        sc.init(P)
        sc.time_int(2)
        snapshot.write(path, cfg, g.download(0), sc.simtime, sc.timestep, sc.last_dt)
        sc.time_int(3)
        want, twant = g.download(0), sc.simtime
    cfg2, P2, meta = snapshot.read(path)
    assert bytes(cfg2) == bytes(cfg)
    with lib.GpuSim(cfg2, 0) as g:
        if setup:
            setup(g)
        sc = driver.SimControl(g, cfg2)
        sc.first_step_dt_limit = dtl
        sc.init(P2, simtime=meta["simtime"])
        sc.timestep, sc.last_dt = meta["timestep"], meta["last_dt"]
        sc.time_int(3)
        assert sc.simtime == twant
        assert np.array_equal(g.download(0), want)
