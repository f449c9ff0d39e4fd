"""What the host grants this process: logical CPUs, affinity, cgroup CPU quota.  Used to choose how prover threads wait for the GPU
(pk_device_set_host_wait): a container that SEES 256 CPUs under a quota of 16 must not run 24 spinning threads."""
import os


def usable_cores() -> dict:
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
    return info
