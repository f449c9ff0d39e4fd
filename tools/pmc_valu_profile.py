#!/usr/bin/env python3
"""VALU accounting of a proof from one `rocprofv3 --pmc VALUBusy --kernel-trace` run of bench.py (tools/pmc_valu.sh): every dispatch's
VALUBusy (% of its cycles a SIMD's vector ALU was executing) x its own duration (the kernel-trace row with the same Dispatch_Id),
summed per kernel and divided by the number of proofs -> profiles/<prefix>_pmc_valu.json.
usage: pmc_valu_profile.py <gpurun_out/tag> <round-prefix> <proofs in the run> "<workload note>" """
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, prefix, proofs, note = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
sha = hashlib.sha256(open(os.path.join(ROOT, "provekit_amd", "lib", "libprovekit_hip.so"), "rb").read()).hexdigest()[:16]
dur = {}
for f in glob.glob(f"{src}/VALUBusy/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
leaf = []  # (duration, VALUBusy) of every leaf-hash dispatch: the largest launches' figure goes into the leaf-hash summary bench.py reads
for f in glob.glob(f"{src}/VALUBusy/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != "VALUBusy" or r["Dispatch_Id"] not in dur:
            continue
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        if "leaf_hash_kernel" in k:
            leaf.append((dur[r["Dispatch_Id"]], float(r["Counter_Value"])))
        if not any(x in k for x in ("_kernel", "Kernel")) or k.startswith(("at::", "__amd", "modmul_rate", "constmul_rate", "sq_round_rate")):  # the peak probe is not part of a proof
            continue
        d = dur[r["Dispatch_Id"]] * 1e-6
        agg[k][0] += 1
        agg[k][1] += d
        agg[k][2] += d * float(r["Counter_Value"]) / 100.0
kern = {k: {"dispatches": v[0], "dur_ms_per_proof": v[1] / proofs, "valu_busy_ms_per_proof": v[2] / proofs, "VALUBusy_pct": 100.0 * v[2] / v[1] if v[1] else 0.0}
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2])}
out = {"lib_sha16": sha, "workload": note, "proofs": proofs, "kernel_ms_per_proof": sum(v[1] for v in agg.values()) / proofs,
       "valu_busy_ms_per_proof": sum(v[2] for v in agg.values()) / proofs, "kernels": kern}
if leaf:
    top = max(d for d, _ in leaf)
    big = [v for d, v in leaf if d >= 0.9 * top]
    out["leaf_hash_valu_busy_pct_largest_launch"] = sum(big) / len(big)
    lh = os.path.join(ROOT, "profiles", f"{prefix}_pmc_leaf_hash.json")
    if os.path.exists(lh):
        j = json.load(open(lh))
        if j.get("lib_sha16") == sha:
            j["valu_busy_pct_largest_launch"] = out["leaf_hash_valu_busy_pct_largest_launch"]
            j["valu_busy_note"] = f"mean VALUBusy of the {len(big)} longest leaf-hash dispatches (the 2^(m-3)-leaf launches) of the VALUBusy pass (tools/pmc_valu.sh)"
            json.dump(j, open(lh, "w"), indent=1)
path = os.path.join(ROOT, "profiles", f"{prefix}_pmc_valu.json")
json.dump(out, open(path, "w"), indent=1)
print(path, round(out["kernel_ms_per_proof"], 3), round(out["valu_busy_ms_per_proof"], 3))
for k, v in list(kern.items())[:8]:
    print(f"  {k[:48]:50s} {v['valu_busy_ms_per_proof']:.3f} ms  {v['VALUBusy_pct']:.0f} %")
