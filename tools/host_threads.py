#!/usr/bin/env python3
"""Where the host CPU of throughput mode goes: per-thread CPU seconds (/proc/self/task/*/stat) over a window of proofs by C provers,
prover threads against everything else in the process (the HIP runtime's own threads).  usage: tools/host_threads.py [C=16] [block|spin|poll] [proofs per prover=12]"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
conc = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mode = sys.argv[2] if len(sys.argv) > 2 else "block"
per = int(sys.argv[3]) if len(sys.argv) > 3 else 12
import torch  # noqa: E402

torch.cuda.is_available()
import bench  # noqa: E402
import provekit_amd  # noqa: E402
from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for  # noqa: E402

if mode != "spin":
    provekit_amd.Context.set_host_wait(0, mode)
m, m_0 = 21, 20
n_wit = (1 << (m - 1)) - 5
cfg_w, cfg_b = WhirConfig.derive(m), blinding_config_for(m_0)
workers = []
for w in range(conc):
    c = provekit_amd.Context(0)
    r1cs, _, _, nc, n_in = bench.synth_r1cs(c, m_0, n_wit, seed=1234)
    d_z, _ = bench.satisfying_witness(c, r1cs, n_wit, nc, n_in, 99 + w)
    workers.append((c, WhirR1CSScheme(c, r1cs, m, m_0, cfg_w, cfg_b), d_z))
TICK = os.sysconf("SC_CLK_TCK")


def task_times():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{tid}/stat").read()
            comm = f[f.index("(") + 1 : f.rindex(")")]
            rest = f[f.rindex(")") + 2 :].split()
            out[int(tid)] = (comm, (int(rest[11]) + int(rest[12])) / TICK, int(rest[11]) / TICK, int(rest[12]) / TICK)
        except (OSError, ValueError):
            pass
    return out


tids = {}


def work(w, n, seed0):
    tids[threading.get_native_id()] = w
    for i in range(n):
        workers[w][1].prove_nocopy(workers[w][2], seed=seed0 + 1000 * w + i)


def wave(n, seed0):
    ths = [threading.Thread(target=work, args=(w, n, seed0)) for w in range(conc)]
    for t in ths:
        t.start()
    a = task_times() if seed0 else None
    for t in ths:
        t.join()
    return a


wave(2, 0)
t0 = time.perf_counter()
ths = [threading.Thread(target=work, args=(w, per, 1)) for w in range(conc)]
for t in ths:
    t.start()
time.sleep(0.05)
a = task_times()
ta = time.perf_counter()
time.sleep(max(0.5, 0.06 * per - 0.2))
b = task_times()
tb = time.perf_counter()
for t in ths:
    t.join()
wall = time.perf_counter() - t0
win = tb - ta
prov = {"cpu_s": 0.0, "user_s": 0.0, "sys_s": 0.0, "threads": 0}
other = {}
for tid, (comm, tot, u, s_) in b.items():
    if tid not in a:
        continue
    d, du, ds = tot - a[tid][1], u - a[tid][2], s_ - a[tid][3]
    if tid in tids:
        prov["cpu_s"] += d
        prov["user_s"] += du
        prov["sys_s"] += ds
        prov["threads"] += 1
    elif d > 0:
        o = other.setdefault(comm, {"cpu_s": 0.0, "user_s": 0.0, "sys_s": 0.0, "threads": 0})
        o["cpu_s"] += d
        o["user_s"] += du
        o["sys_s"] += ds
        o["threads"] += 1
rate = conc * per / wall
print(json.dumps({"provers": conc, "host_wait": mode, "proofs_per_s": round(rate, 1), "window_s": round(win, 3),
                  "prover_threads": {k: round(v, 3) if isinstance(v, float) else v for k, v in prov.items()},
                  "prover_threads_cores_busy": round(prov["cpu_s"] / win, 2),
                  "prover_thread_cpu_ms_per_proof": round(1e3 * prov["cpu_s"] / (rate * win), 2),
                  "other_threads": {k: {kk: round(vv, 3) if isinstance(vv, float) else vv for kk, vv in v.items()} for k, v in sorted(other.items(), key=lambda kv: -kv[1]["cpu_s"])},
                  "other_threads_cores_busy": round(sum(v["cpu_s"] for v in other.values()) / win, 2)}))
