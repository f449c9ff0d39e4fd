#!/bin/bash
# A/B of the register NTT's elements per lane (PK_NTT_LE = 3: eight, two waves per SIMD; 2: four, four waves per SIMD), same binary,
# alternating, same box.  One JSON line per run.  HISTORICAL: the switch existed only in the intermediate build of round 6 that produced
# profiles/r06_ntt_le_ab.jsonl (commit 32f6759's parent state); the tree keeps four per lane only, so today both arms run the same kernel.
Q='--no-cpu-baseline --no-commit-probe --size-classes= --no-h2d-probe --no-latency-pass'
for i in $(seq 1 ${ROUNDS:-2}); do for LE in ${LES:-3 2}; do
  export PK_NTT_LE=$LE
  python tools/ntt_ab.py 2>/dev/null | LE=$LE python -c 'import json,sys,os; d=json.loads(sys.stdin.read()); d["le"]=int(os.environ["LE"]); print(json.dumps(d))'
  [ -n "$SKIP_BENCH" ] && continue
  timeout 300 python bench.py --steps 12 --warmup 3 $Q 2>/dev/null | LE=$LE python -c 'import json,sys,os; d=json.loads(sys.stdin.read()); print(json.dumps({"le": int(os.environ["LE"]), "m21_proofs_per_s": round(d["value"], 2)}))'
  for m in 23 25; do
    timeout 300 python bench.py --size-class-probe $m 2>/dev/null | LE=$LE M=$m python -c 'import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({"le": int(os.environ["LE"]), "m": int(os.environ["M"]), "proofs_per_s": round(d["proofs_per_s"],2), "single_proof_ms": round(d["single_proof_ms"],2), "provers": d["provers"]}))'
  done
  timeout 300 python bench.py --workload commit --m 26 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | LE=$LE python -c 'import json,sys,os; d=json.loads(sys.stdin.read()); print(json.dumps({"le": int(os.environ["LE"]), "commit26_ms": round(d["ms_per_step"], 3), "root": d.get("config", {}).get("root", d.get("root"))[:16]}))'
done; done
