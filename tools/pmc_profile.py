#!/usr/bin/env python3
"""Turns the per-kernel FETCH_SIZE / WRITE_SIZE sums of tools/pmc.sh (gpurun_out/<tag>/pmc_summary.json) into the committed
summaries bench.py reads (profiles/<round>_pmc_leaf_hash.json) and DESIGN.md quotes (profiles/<round>_pmc_ntt.json).
Every summary records the sha256 of the library binary it was taken with: bench.py reports `roofline.traffic` only when
that matches the binary it is running (a stale number is reported as null instead).
usage: pmc_profile.py <gpurun_out/tag> <round-prefix> prove <m> | commit <log2-size> <steps+warmup>"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, prefix, kind = sys.argv[1], sys.argv[2], sys.argv[3]
S = json.load(open(os.path.join(src, "pmc_summary.json")))
sha = hashlib.sha256(open(os.path.join(ROOT, "provekit_amd", "lib", "libprovekit_hip.so"), "rb").read()).hexdigest()[:16]
CORR = ("FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at 64 B: MI355X_MICROARCH.md, HBM), WRITE_SIZE as reported; both in KiB "
        "per dispatch, separate --pmc passes (tools/pmc.sh)")


def per_kernel(match):
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, v in S[c].items():
            if match in k:
                e = out.setdefault(k, {"dispatches": v["dispatches"], "fetch_kib": 0.0, "write_kib": 0.0})
                e["fetch_kib" if c == "FETCH_SIZE" else "write_kib"] = v["sum"]
    return out


if kind == "prove":
    m = int(sys.argv[4])
    ks = per_kernel("leaf_hash_kernel")
    n = sum(v["dispatches"] for v in ks.values())
    fetch, write = sum(v["fetch_kib"] for v in ks.values()), sum(v["write_kib"] for v in ks.values())
    out = {"kernel": "leaf_hash_kernel (all launches of a proof, as bench.py's roofline averages them)", "lib_sha16": sha, "m": m,
           "workload": f"bench.py prove, m={m}, --concurrency 1 (tools/pmc.sh)", "dispatches": n, "kernels": ks,
           "fetch_size_kib_per_dispatch": fetch / n, "write_size_kib_per_dispatch": write / n, "correction": CORR,
           "traffic_bytes_per_launch": (2.0 * fetch + write) * 1024.0 / n}
    path = os.path.join(ROOT, "profiles", f"{prefix}_pmc_leaf_hash.json" if m == 21 else f"{prefix}_m{m}_pmc_leaf_hash.json")
    # the RS-encode kernels of the same run (BASELINE configs[2] asks for the NTT's HBM figures at the sha256 size class too)
    nk = {**per_kernel("ntt8_pass_kernel"), **per_kernel("deinterleave_kernel"), **per_kernel("ntt_pass_kernel")}
    if nk:
        proofs = max(1, round(n / 7.0))  # 7 leaf-hash launches per proof (5 witness-WHIR trees + 2 blinding trees)
        elems = 0
        for n_vars, batch, rounds in ((m, 2, m // 4 - 1), (None, 2, None)):
            if n_vars is None:
                break
            rows = 1 << (n_vars + 1 - 4)
            elems += rows * 16 * batch
            for _ in range(rounds):
                rows >>= 1
                elems += rows * 16
        nf, nw = sum(v["fetch_kib"] for v in nk.values()), sum(v["write_kib"] for v in nk.values())
        traffic = (2.0 * nf + nw) * 1024.0 / proofs
        ntt = {"kernels": "deinterleave_kernel + ntt8_pass_kernel + ntt_pass_kernel of one proof (all RS-encodes)", "lib_sha16": sha, "m": m,
               "workload": f"bench.py prove, m={m}, --concurrency 1 (tools/pmc.sh)", "proofs_in_the_trace": proofs, "per_kernel": nk,
               "correction": CORR, "codeword_elements_per_proof_witness_whir": elems, "algorithmic_bytes_per_proof": 64.0 * elems,
               "traffic_bytes_per_proof": traffic, "traffic_over_algorithmic": traffic / (64.0 * elems)}
        json.dump(ntt, open(os.path.join(ROOT, "profiles", f"{prefix}_pmc_ntt_m{m}.json"), "w"), indent=1)
else:
    log2, runs = int(sys.argv[4]), int(sys.argv[5])
    ks = {**per_kernel("ntt8_pass_kernel"), **per_kernel("deinterleave_kernel"), **per_kernel("leaf_hash_kernel")}
    rows, cols = 1 << (log2 + 1 - 4), 32
    elems = rows * cols
    ntt = {k: v for k, v in ks.items() if "ntt8" in k or "deinterleave" in k}
    fetch, write = sum(v["fetch_kib"] for v in ntt.values()), sum(v["write_kib"] for v in ntt.values())
    traffic = (2.0 * fetch + write) * 1024.0 / runs
    out = {"kernels": "deinterleave_kernel + ntt8_pass_kernel of one batch-2 RS-encode", "lib_sha16": sha, "log2_size": log2,
           "workload": f"bench.py --workload commit --log2-size {log2} on one GPU, {runs} commits (tools/pmc.sh)", "per_kernel": ks,
           "correction": CORR, "codeword_elements": elems, "algorithmic_bytes_per_encode": 64.0 * elems,
           "traffic_bytes_per_encode": traffic, "traffic_over_algorithmic": traffic / (64.0 * elems)}
    path = os.path.join(ROOT, "profiles", f"{prefix}_pmc_ntt.json")
json.dump(out, open(path, "w"), indent=1)
print(path, json.dumps({k: v for k, v in out.items() if not isinstance(v, dict)})[:600])
