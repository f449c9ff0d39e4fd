#!/bin/bash
# A/B of NTT builds: alternating, same box; libs = tools/_ab/*.so named on the command line + the tree's build.
# One JSON line per run: isolated rs_encode timings + digests, the bench headline (m = 21), the size classes (23, 25) and the 2^26 commit.
# LIBS: the builds to compare, e.g. a copy of an older commit's libprovekit_hip.so under tools/ab/ (git-ignored as *.so, but it travels with gpurun) and the tree's
LIBS="${LIBS:-provekit_amd/lib/libprovekit_hip.so}"
Q='--no-cpu-baseline --no-commit-probe --size-classes= --no-h2d-probe --no-latency-pass'
for i in $(seq 1 ${ROUNDS:-2}); do for L in $LIBS; do
  export PK_LIB_PATH=$PWD/$L
  python tools/ntt_ab.py 2>/dev/null
  [ -n "$SKIP_BENCH" ] && continue
  timeout 300 python bench.py --steps 12 --warmup 3 $Q 2>/dev/null | L=$L python -c 'import json,sys,os; d=json.loads(sys.stdin.read()); print(json.dumps({"lib": os.path.basename(os.environ["L"]), "m21_proofs_per_s": round(d["value"], 2)}))'
  for m in 23 25; do
    timeout 300 python bench.py --size-class-probe $m 2>/dev/null | L=$L M=$m python -c 'import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({"lib": os.path.basename(os.environ["L"]), "m": int(os.environ["M"]), "probe": d}))'
  done
  timeout 300 python bench.py --workload commit --m 26 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | L=$L python -c 'import json,sys,os; d=json.loads(sys.stdin.read()); print(json.dumps({"lib": os.path.basename(os.environ["L"]), "commit26_ms": round(d["ms_per_step"], 3), "root": d.get("config", {}).get("root", d.get("root"))}))'
done; done
