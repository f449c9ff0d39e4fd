#!/bin/bash
# A/B of VERDICT r04 weak #13: hipEvent pairs around worker 0's launches inside the timed region (default) against none
# (PK_BENCH_NO_TIMED_PROFILE=1), alternating, same box, same binary.  One JSON line per run.
Q='--steps 20 --warmup 5 --no-cpu-baseline --no-commit-probe --size-classes= --no-h2d-probe --no-latency-pass'
for i in 1 2 3; do for off in 0 1; do
  PK_BENCH_NO_TIMED_PROFILE=$off python bench.py $Q 2>/dev/null | OFF=$off python -c 'import json,sys,os; d=json.loads(sys.stdin.read()); print(json.dumps({"profiling_in_timed_region": os.environ["OFF"] == "0", "proofs_per_s": round(d["value"], 2)}))'
done; done
