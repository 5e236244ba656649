"""Host facts for sizing thread teams (python twin of host/glc_cpus.h)."""
import os


def effective_cpus():
    """CPUs this process can really use: min(online CPUs, affinity mask, cgroup CPU quota, OMP_NUM_THREADS).  A GPU box shows
    all of its hardware threads but grants a job a share; OpenMP teams larger than the share spend their time being throttled."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, -(-int(txt[0]) // int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0 and period > 0:
                    n = min(n, max(1, -(-quota // period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    env = os.environ.get("GLO_THREADS") or os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit() and int(env) > 0:
        n = min(n, int(env))
    return max(1, n)
