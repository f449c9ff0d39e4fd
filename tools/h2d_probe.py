#!/usr/bin/env python3
"""PCIe-inclusive rate of the bench workload: witness resident vs uploaded before every proof from pageable vs pinned host memory
(16 provers, dynamic hand-out as in bench.py); "sleep" = the prover thread idle for 1.6 ms instead, nothing copied.  GPU box.  usage: h2d_probe.py [waves]"""
import itertools, json, os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import provekit_amd
from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for

waves = int(sys.argv[1]) if len(sys.argv) > 1 else 10
m, conc = 21, 16
m_0, n_wit = m - 1, (1 << (m - 1)) - 5
cfg_w, cfg_b = WhirConfig.derive(m), blinding_config_for(m_0)
workers = []
for w in range(conc):
    c = provekit_amd.Context(0)
    r1cs, mats, interner, nc, n_in = bench.synth_r1cs(c, m_0, n_wit, seed=1234)
    d_z, z_host = bench.satisfying_witness(c, r1cs, n_wit, nc, n_in, 99 + 1000 * w)
    z_host = np.ascontiguousarray(z_host)
    pinned = torch.from_numpy(z_host.view(np.int64).copy()).pin_memory()
    workers.append((c, WhirR1CSScheme(c, r1cs, m, m_0, cfg_w, cfg_b), d_z, z_host, pinned))


def run(count, mode, first_seed):
    nxt, lock = itertools.count(), threading.Lock()

    def work(w):
        c, prover, d_z, z_host, pinned = workers[w]
        while True:
            with lock:
                i = next(nxt)
            if i >= count:
                return
            if mode == "pageable":
                c.upload_into(d_z.ptr, z_host)
            elif mode == "pinned":
                c._check(provekit_amd._lib.lib.pk_memcpy_h2d(c.handle, d_z.ptr, pinned.data_ptr(), pinned.numel() * 8))
            elif mode == "sleep":  # the prover thread idle for about as long as an upload takes, nothing copied
                time.sleep(0.0016)
            prover.prove_nocopy(d_z, seed=first_seed + i)

    ths = [threading.Thread(target=work, args=(w,)) for w in range(conc)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    return count / (time.perf_counter() - t0)


run(2 * conc, "resident", 0)
res = {"resident": [], "pageable": [], "pinned": [], "sleep": []}
for rep in range(3):
    for mode in res:
        run(conc, mode, 1000)
        res[mode].append(round(run(waves * conc, mode, 5000 + 100 * rep), 1))
print(json.dumps({"m": m, "provers": conc, "proofs_per_s": res}))
