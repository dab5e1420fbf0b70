"""Host core budget: min(scheduler affinity, cgroup CPU quota).  The GPU boxes expose 256 logical CPUs but cap the
container at 16 via cgroup cpu.max; a thread pool sized from os.cpu_count() there is throttled into the ground."""
import math
import os


def usable_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, math.floor(int(parts[0]) / int(parts[1]))))
            else:
                quota = int(parts[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, quota // int(g.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)
