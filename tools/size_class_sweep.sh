#!/bin/bash
# provers in flight per size class (m = 23, 25), arena sized from pk_scheme_arena_bytes; one JSON line per run, fresh process each
for spec in ${SPECS:-"25 3 spin" "25 3 poll" "25 4 spin" "25 4 poll" "25 6 poll" "25 6 spin" "25 8 poll" "23 14 poll" "23 16 poll" "23 18 poll" "23 20 poll" "23 24 poll" "25 4 poll" "25 6 poll" "25 3 spin"}; do
  set -- $spec
  PK_BENCH_SIZE_CLASS_PROVERS=$2 PK_BENCH_SIZE_CLASS_WAIT=$3 timeout 400 python bench.py --size-class-probe $1 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({k: d[k] for k in ("m","provers","host_wait","proofs_per_s","single_proof_ms","arena_bytes_per_prover")}))'
done
