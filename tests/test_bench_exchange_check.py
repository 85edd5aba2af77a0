"""bench.py's end-of-run exchange check for N > 1 (the transport between two different GPUs could not be exercised in
development): its two pure parts, on slabs cut from a periodic global array -- consistent ghosts pass, one flipped
bit in one ghost plane, a stale face or a NaN is named."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def _slabs(world, periodic, nvar=3, nz=8, ny=5, nx=6, nb=2, seed=3):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((nvar, nz, ny + 2 * nb, nx + 2 * nb))
    nzl = nz // world
    out = []
    for r in range(world):
        A = np.zeros((nvar, nzl + 2 * nb, ny + 2 * nb, nx + 2 * nb))
        for k in range(-nb, nzl + nb):
            kg = r * nzl + k
            if periodic:
                A[:, k + nb] = G[:, kg % nz]
            elif 0 <= kg < nz:
                A[:, k + nb] = G[:, kg]
            else:
                A[:, k + nb] = -7.0   # a physical boundary's ghosts: whatever
        out.append(A)
    return out, nb, nzl


def test_consistent_slabs_pass():
    for world in (2, 4):
        for periodic in (True, False):
            slabs, nb, nzl = _slabs(world, periodic)
            every = [bench.exchange_digests(A, nb, nzl) for A in slabs]
            assert bench.compare_exchange(every, periodic) == []


def test_one_flipped_bit_in_a_ghost_plane_is_found():
    slabs, nb, nzl = _slabs(4, True)
    A = slabs[2]
    v = A[1, 0, 3, 4:5].view(np.uint64)
    v ^= np.uint64(1)
    every = [bench.exchange_digests(A, nb, nzl) for A in slabs]
    bad = bench.compare_exchange(every, True)
    assert bad == ["rank 2 lower ghosts != rank 1 top planes"]


def test_stale_upper_face_and_nan_are_named():
    slabs, nb, nzl = _slabs(2, False)
    slabs[0][:, nzl + nb:] += 1.0
    slabs[1][0, nb + 1, 2, 2] = np.nan
    every = [bench.exchange_digests(A, nb, nzl) for A in slabs]
    bad = bench.compare_exchange(every, False)
    assert "rank 0 upper ghosts != rank 1 bottom planes" in bad
    assert "rank 1 holds non-finite values" in bad
    # the physical faces of the end ranks are not compared
    assert not any("rank 0 lower" in b or "rank 1 upper" in b for b in bad)
