"""pion_amd -- MI355X-native flux-update hot path of PION behind a C-ABI.

The product is pion_amd/csrc/libpion_gpu.so (hand-written HIP for gfx950,
include/pion_gpu.h).  This package only holds the thin host-side glue:
abi.py (ctypes view of the header), lib.py (handle wrapper), problems.py
(initial conditions of the reference's test problems), driver.py (the
sim_control-shaped time loop, single- and multi-GPU).
"""
from . import abi  # noqa: F401
