"""How many cores the CPU oracle may really use.  TEST INFRASTRUCTURE (oracle/)."""
import os


def usable_cores() -> dict:
    """How many cores this process may really use: the logical CPUs it sees, its affinity mask, and the cgroup CPU quota (a container
    that sees 256 CPUs under a quota of 16 runs SLOWER with 128 OpenMP threads than with 16: the rest is throttled)."""
    info = {"logical_cpus": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "cgroup_cpu_quota": None}
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            info["cgroup_cpu_quota"] = int(quota) / int(period)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            if q > 0:
                info["cgroup_cpu_quota"] = q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError):
            pass
    n = min(info["logical_cpus"], info["affinity"])
    if info["cgroup_cpu_quota"]:
        n = min(n, max(1, int(info["cgroup_cpu_quota"])))
    info["usable"] = n
    try:
        info["model"] = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except (OSError, IndexError):
        pass
    return info
