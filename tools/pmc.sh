#!/bin/bash
# Run on the GPU box: HBM traffic counters for the dominant kernel, one --pmc pass per counter
# (MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE do not fit one pass; FETCH_SIZE reads 1/2 on gfx950).
# usage: tools/pmc.sh <tag> [bench args...]
set -u
TAG=${1:-pmc}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
mkdir -p "$R/gpurun_out/$TAG"
for C in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && rocprofv3 --pmc $C --kernel-trace -f csv -d "$R/gpurun_out/$TAG/$C" -o pmc -- python "$R/bench.py" --no-cpu-baseline "$@" > "$R/gpurun_out/$TAG/$C.log" 2>&1
done
python3 - "$R/gpurun_out/$TAG" <<'PY'
import csv, glob, sys, collections, json
root = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"{root}/{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c:
                continue
            k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            agg[k][0] += 1
            agg[k][1] += float(row["Counter_Value"])
    out[c] = {k: {"dispatches": v[0], "sum": v[1], "per_dispatch": v[1] / max(v[0], 1)} for k, v in agg.items()}
json.dump(out, open(f"{root}/pmc_summary.json", "w"), indent=1)
for c in out:
    for k, v in sorted(out[c].items(), key=lambda kv: -kv[1]["sum"])[:8]:
        print(c, k[:70], v["dispatches"], round(v["per_dispatch"], 1))
PY
find "$R/gpurun_out/$TAG" -name '*.csv' -size +2M -delete
