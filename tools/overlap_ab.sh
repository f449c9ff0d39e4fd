#!/bin/bash
# A/B: the blinding commitment + external rows on a side stream in PLAIN mode too (PK_BLINDING_OVERLAP=1) against the default (latency mode only).
# HISTORICAL: the switch was a one-line development patch of prover.hip (overlap_blinding) that produced profiles/r06_side_stream_plain_mode_ab.jsonl
# and was not kept (the library reads no such variable).
Q='--no-cpu-baseline --no-commit-probe --size-classes= --no-h2d-probe'
for i in $(seq 1 ${ROUNDS:-3}); do for ov in 0 1; do
  if [ $ov = 1 ]; then export PK_BLINDING_OVERLAP=1; else unset PK_BLINDING_OVERLAP; fi
  timeout 300 python bench.py --steps 12 --warmup 3 $Q 2>/dev/null | OV=$ov python -c 'import json,sys,os; d=json.loads(sys.stdin.read()); s=d["single_stream"]; print(json.dumps({"side_stream_in_plain_mode": os.environ["OV"]=="1", "proofs_per_s": round(d["value"],2), "single_ms": round(s["ms_per_proof"],3), "single_latency_mode_ms": round(s.get("latency_mode_ms_per_proof") or 0,3)}))'
  timeout 300 python bench.py --size-class-probe 23 2>/dev/null | OV=$ov python -c 'import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({"side_stream_in_plain_mode": os.environ["OV"]=="1", "m": 23, "proofs_per_s": round(d["proofs_per_s"],2), "single_proof_ms": round(d["single_proof_ms"],2)}))'
done; done
