#!/bin/bash
# Run on the GPU box: where a kernel's wave-cycles go (SQ counters, one --pmc pass per group), for any command.
# usage: tools/pmc_sq.sh <tag> <kernel-name-substring> -- <command...>     (PK_LIB_PATH etc. from the environment)
set -u
TAG=$1; PAT=$2; shift 3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
mkdir -p "$R/gpurun_out/$TAG"
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
G2="SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"
G3="GRBM_GUI_ACTIVE GRBM_COUNT SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_IFETCH"
i=0
for G in "${PMC_G1:-$G1}" "${PMC_G2:-$G2}" "${PMC_G3:-$G3}"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $G --kernel-trace -f csv -d "$R/gpurun_out/$TAG/g$i" -o pmc -- "$@" > "$R/gpurun_out/$TAG/g$i.log" 2>&1)
done
python3 - "$R/gpurun_out/$TAG" "$PAT" <<'PY'
import csv, glob, sys, collections, json
root, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(f"{root}/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        if pat not in k:
            continue
        a = agg[k][row["Counter_Name"]]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
out = {k: {c: round(v[1] / max(v[0], 1), 1) for c, v in cs.items()} | {"dispatches": max(v[0] for v in cs.values())} for k, cs in agg.items()}
json.dump(out, open(f"{root}/pmc_sq_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find "$R/gpurun_out/$TAG" -name '*.csv' -size +2M -delete
