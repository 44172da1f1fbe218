"""Host-CPU budget of the training process.

The GPU path needs ONE busy host thread (it enqueues hipGraph launches and waits on the stream).  A GPU container
typically SEES every logical CPU of the node (256) but is GRANTED a cgroup quota (e.g. ``cpu.max = 1600000 100000`` = 16
CPUs).  OpenMP / BLAS pools size themselves by the visible count and spin after every parallel region; under the quota
that exhausts the cgroup's budget within a scheduling period and the kernel throttles the WHOLE process for tens of
milliseconds -- the GPU then sits idle waiting for the next graph launch (measured: 200-step runs at 2-4x their normal
time, a 20-step run at 1.9 ms/step instead of 0.1; ``profiles/r02_host_throttling.txt``).

Import this module and call :func:`limit_host_threads` BEFORE numpy / torch are imported (no heavy imports here).
"""
import os


def cpu_budget():
    """CPUs this process may use: the cgroup (v2 or v1) CPU quota when there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0 and per > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def limit_host_threads(max_threads=4):
    """Size the OpenMP / BLAS pools of this process for a GPU-driving process: at most ``max_threads`` and at most a
    quarter of this rank's share of the CPU budget.  Respects values the user already exported.  Returns the thread count chosen."""
    try:        # one process per GPU: the ranks of this node share the budget
        ranks = max(1, int(os.environ.get('LOCAL_WORLD_SIZE') or os.environ.get('WORLD_SIZE') or 1))
    except ValueError:
        ranks = 1
    nt = max(1, min(int(max_threads), cpu_budget() // (4 * ranks) or 1))
    for k in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'NUMEXPR_NUM_THREADS'):
        os.environ.setdefault(k, str(nt))
    return int(os.environ['OMP_NUM_THREADS'])
