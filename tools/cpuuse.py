#!/usr/bin/env python3
"""Host CPU a throughput run burns: wall and CPU seconds of `bench.py --concurrency C` (children included), i.e. the average number of
cores busy -- against the cgroup CPU quota of the box (oracle/hostcores.py).  usage: tools/cpuuse.py C [extra bench args...]"""
import json
import os
import resource
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from hostcores import usable_cores  # noqa: E402

c = sys.argv[1]
t = time.time()
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--concurrency", c, "--steps", "12", "--no-commit-probe", "--size-classes", "",
                      "--no-h2d-probe", "--no-latency-pass"] + sys.argv[2:], capture_output=True, text=True)
w = time.time() - t
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
thr = {}
try:
    for l in open("/sys/fs/cgroup/cpu.stat"):
        k, v = l.split()
        if k in ("nr_periods", "nr_throttled", "throttled_usec"):
            thr[k] = int(v)
except OSError:
    pass
print(json.dumps({"provers": int(c), "proofs_per_s": round(d["value"], 1), "wall_s": round(w, 1), "cpu_s": round(ru.ru_utime + ru.ru_stime, 1),
                  "user_s": round(ru.ru_utime, 1), "sys_s": round(ru.ru_stime, 1), "avg_cores_busy_incl_setup": round((ru.ru_utime + ru.ru_stime) / w, 1),
                  "usable_cores": usable_cores()["usable"], "cgroup_cpu_stat_cumulative": thr, "env": {k: os.environ[k] for k in os.environ if k.startswith("PK_")}}))
