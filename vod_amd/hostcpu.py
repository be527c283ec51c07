"""How many CPUs this process may really use, and a cap on the thread pools that would otherwise assume the whole machine.

A container commonly sees every core of the host (`os.cpu_count()` = 256 on the MI355X boxes of this pool) while its cgroup
grants a fraction (16 CPUs).  Thread pools sized by the visible count - OpenMP inside torch's CPU ops, MKL / OpenBLAS - then
spin on cores they are not allowed to burn: the CFS quota of the period is gone in milliseconds and EVERY thread of the cgroup is
frozen until the next period.  Measured on this pool: a worker-group request (two torch CPU ops above the parallel grain size on its
path) took 200 ms instead of 1.1 ms.  The server processes therefore cap their pools at start-up (`limit_cpu_threads`).
"""
from __future__ import annotations

import math
import os


def usable_cpus() -> int:
    """Scheduler affinity capped by the cgroup CPU quota (v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = period = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota, period = int(q), int(p)
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError):
            quota = None
    if quota and period and quota > 0:
        n = min(n, max(1, math.ceil(quota / period)))
    return max(1, n)


def limit_cpu_threads(limit: int | None = None, export: bool = True) -> int:
    """Cap torch's intra-op pool (and, for child processes, the OpenMP / MKL / OpenBLAS environment defaults) at the usable CPUs.
    A value the user already exported wins.  Returns the cap."""
    n = usable_cpus() if limit is None else max(1, int(limit))
    if export:
        for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
            os.environ.setdefault(var, str(n))
    try:
        import torch

        if torch.get_num_threads() > n:
            torch.set_num_threads(n)
    except ImportError:  # pragma: no cover - the owner process of a worker group never imports torch
        pass
    return n
