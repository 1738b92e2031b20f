"""What the host side may use.

``os.cpu_count()`` (and torch's default intra-op thread count, which follows it) reports the
cores of the MACHINE; a container is often given a CPU-time quota far below that (the MI355X boxes
here: 256 hardware threads visible, ``cpu.max`` = 16 CPUs).  A thread pool sized for the machine
then spends its time being throttled by the scheduler: the float64 CPU oracle ran 7 x slower with
torch's default 128 threads than with 16, and not at all with 256 (tools/oracle_threads.py,
round 4) -- which is also what stalled round 3's two-process GPU tests (two ranks + the test
process = 384 busy threads on 16 CPUs).
"""

import os

__all__ = ['usable_cpus', 'limit_host_threads']


def _cgroup_quota():
    """CPUs granted by the cgroup CPU controller (v2 ``cpu.max``, v1 ``cfs_quota_us``) or None."""
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()[:2]
        if quota != 'max':
            return float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
            quota = float(f.read())
        with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
            period = float(f.read())
        if quota > 0 and period > 0:
            return quota / period
    except (OSError, ValueError):
        pass
    return None


def usable_cpus():
    """Number of CPUs this process can actually keep busy: min(affinity mask, cgroup quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = _cgroup_quota()
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def limit_host_threads(share=1, cap=None):
    """Size torch's intra-op pool (and what child processes inherit through OMP_NUM_THREADS /
    MKL_NUM_THREADS) to ``usable_cpus() // share``; -> the thread count set."""
    import torch
    n = max(1, usable_cpus() // max(1, int(share)))
    if cap is not None:
        n = min(n, int(cap))
    torch.set_num_threads(n)
    os.environ['OMP_NUM_THREADS'] = str(n)
    os.environ['MKL_NUM_THREADS'] = str(n)
    return n
