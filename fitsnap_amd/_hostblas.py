"""Threads of the numpy / scipy BLAS underneath the K x K host algebra of the solver classes that still use it (ARD's
``pinvh``, ANL's ``pinv``).

The statistics come from the GPU; what is left on the host is an eigen- or singular-value decomposition of a K x K matrix.
OpenBLAS / MKL size their pools by the CPUs they SEE: on the MI355X boxes that is 256 logical CPUs behind a cgroup quota of 16,
and a 128 x 128 ``eigh`` then takes 12 ms instead of 0.6 (six of them were 72 of the 89 ms of an ARD fit at 10^6 x 128,
``scripts/consumer_fit_survey.py``).  ``blas_threads(K)`` limits the pools for the duration of such a call: one thread per 128
columns, never more than the CPUs this process may use.  A no-op without ``threadpoolctl`` (a dependency of scikit-learn, which
the reference's own ARD solver needs)."""
from __future__ import annotations

import contextlib
import os

_controller = None
_budget = None


def cpu_budget():
    """CPUs this process may use: its affinity mask, cut by a cgroup CPU quota (cpu.max / cfs_quota_us)."""
    global _budget
    if _budget is None:
        try:
            n = len(os.sched_getaffinity(0))
        except Exception:
            n = os.cpu_count() or 1
        quota = None
        try:
            with open("/sys/fs/cgroup/cpu.max") as f:
                q, p = f.read().split()[:2]
            quota = None if q == "max" else float(q) / float(p)
        except Exception:
            try:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                    q = float(f.read())
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    p = float(f.read())
                quota = q / p if q > 0 and p > 0 else None
            except Exception:
                quota = None
        if quota is not None:
            n = min(n, max(1, int(quota)))
        _budget = max(1, n)
    return _budget


def blas_threads(K):
    """Context manager: BLAS / OpenMP pools limited to what a K x K decomposition can use."""
    global _controller
    try:
        if _controller is None:
            from threadpoolctl import ThreadpoolController
            _controller = ThreadpoolController()        # (the libraries are looked up once: ~1 ms)
        return _controller.limit(limits=max(1, min(cpu_budget(), int(K) // 128)))
    except Exception:
        return contextlib.nullcontext()
