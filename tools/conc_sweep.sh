#!/bin/bash
# proofs/s against the number of provers in flight and the hardware-queue count (throughput mode); one line per setting
for c in ${CONCS:-16 24 32}; do for q in ${QUEUES:-24 40}; do
  GPU_MAX_HW_QUEUES=$q python bench.py --concurrency $c --steps 8 --no-cpu-baseline --no-commit-probe --size-classes "" --no-h2d-probe --no-latency-pass 2>/dev/null \
    | C=$c Q=$q python -c 'import json,sys,os; d=json.loads(sys.stdin.read()); print(json.dumps({"provers": int(os.environ["C"]), "hw_queues": int(os.environ["Q"]), "proofs_per_s": round(d["value"],1)}))'
done; done
