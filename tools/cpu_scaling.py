#!/usr/bin/env python3
"""How the host cores of the box scale for the CPU baseline: compress_many (pure ALU, no memory) at 1..all OpenMP threads, plus
what the container may use (nproc, cgroup quota, affinity).  usage: tools/cpu_scaling.py  -> one JSON object"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests")]
import oracle_lib as o  # noqa: E402

out = {"os_cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        out[f] = open(f).read().strip()
    except OSError:
        pass
try:
    out["model"] = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
except Exception:
    pass
n = 1 << 20
msgs = np.random.default_rng(0).integers(0, 256, size=64 * n, dtype=np.uint8).tobytes()
rates = {}
for t in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if t > 2 * out["os_cpu_count"]:
        break
    o.L.pko_set_num_threads(t)
    nn = n if t >= 8 else n // 8
    o.compress_many(msgs[: 64 * (nn // 8)])
    t0 = time.perf_counter()
    o.compress_many(msgs[: 64 * nn])
    dt = time.perf_counter() - t0
    rates[str(t)] = round(nn / dt / 1e6, 3)
out["compress_M_per_s_by_threads"] = rates
print(json.dumps(out))
