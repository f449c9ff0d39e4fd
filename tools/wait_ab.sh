#!/bin/bash
# resident-witness throughput: provers x host wait mode, alternating, 3 rounds (one box, one binary).  One JSON line per run.
Q='--steps 16 --warmup 4 --no-cpu-baseline --no-commit-probe --size-classes= --no-h2d-probe --no-latency-pass'
for i in $(seq 1 ${ROUNDS:-3}); do for c in ${CONCS:-16 20}; do for w in ${WAITS:-spin block}; do
  timeout 200 python bench.py --concurrency $c --host-wait $w $Q 2>/dev/null | C=$c W=$w python -c 'import json,sys,os; d=json.loads(sys.stdin.read()); print(json.dumps({"provers": int(os.environ["C"]), "host_wait": os.environ["W"], "proofs_per_s": round(d["value"],1)}))'
done; done; done
