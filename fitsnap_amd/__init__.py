"""fitsnap_amd — MI355X (gfx950) linear-fit hot path for FitSNAP.

Hand-written HIP kernels (fused mask x weight x fp64-MFMA normal equations, row
weighting, streaming GEMV) behind a C ABI (include/fsnap_hip.h, libfsnap_hip.so), and a
Python host layer that mirrors the reference's Solver / ParallelTools plugin surface
(fitsnap3lib/solvers, fitsnap3lib/parallel_tools.py).  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"

from . import build  # noqa: F401
