#!/bin/bash
# Run on the GPU box: VALU issue utilisation of the hot kernels, one --pmc pass per derived metric
# (VALUBusy = % of cycles a SIMD's vector ALU is executing; counters do not share a pass with other collectors).
# usage: tools/pmc_valu.sh <tag> [bench args...]
set -u
TAG=${1:-pmcv}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
mkdir -p "$R/gpurun_out/$TAG"
for C in ${PMC_COUNTERS:-VALUBusy VALUUtilization SIMD_UTILIZATION}; do
  cd /tmp && rocprofv3 --pmc $C --kernel-trace -f csv -d "$R/gpurun_out/$TAG/$C" -o pmc -- python "$R/bench.py" --no-cpu-baseline "$@" > "$R/gpurun_out/$TAG/$C.log" 2>&1
done
python3 - "$R/gpurun_out/$TAG" <<'PY'
import csv, glob, sys, collections, json, os
root = sys.argv[1]
out = {}
for d in sorted(os.listdir(root)):
    files = glob.glob(f"{root}/{d}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for f in files:
        rows = list(csv.DictReader(open(f)))
        for row in rows:
            if row.get("Counter_Name") != d:
                continue
            k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            g = int(row.get("Grid_Size", 0) or 0)
            v = float(row["Counter_Value"])
            agg[k][0] += 1
            agg[k][1] += v
            agg[k][2] += v * g  # weight by launch size: big launches dominate the time
    out[d] = {k: {"dispatches": v[0], "mean": v[1] / max(v[0], 1)} for k, v in agg.items()}
json.dump(out, open(f"{root}/pmc_valu_summary.json", "w"), indent=1)
for c in out:
    for k, v in sorted(out[c].items(), key=lambda kv: -kv[1]["dispatches"])[:40]:
        if any(x in k for x in ("leaf_hash", "ntt8", "compress", "sumcheck_cubic_kernel", "modmul", "merkle")):
            print(c, k[:50], v["dispatches"], round(v["mean"], 2))
PY
find "$R/gpurun_out/$TAG" -name '*.csv' -size +2M -delete
