#!/bin/bash
# sclk / socket power sampled twice a second while bench.py runs 150 waves of 16 proofs (GPU box)
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/power_trace.txt &
SM=$!
python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-commit-probe --no-h2d-probe --no-latency-pass --size-classes= 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms/step', d['ms_per_step'])"
wait $SM
awk '{print $0}' gpurun_out/power_trace.txt | sed 's/GPU\[0\]//g; s/\t//g' | cut -c1-150
