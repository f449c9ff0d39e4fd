#!/bin/bash
# Round-6 profile set on the GPU box (one call): bench line, rocprofv3 kernel stats (16 provers / one at a time), PMC HBM traffic at
# m = 21 and m = 23 and for the 2^26 commit, VALUBusy.  Everything lands in gpurun_out/; tools/r06_collect.sh copies the summaries.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
Q='--no-commit-probe --no-h2d-probe --no-latency-pass --size-classes='
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
bash tools/profile.sh r06_conc20 --steps 20 --warmup 5 $Q > gpurun_out/r06_conc20.log 2>&1
KEEP_TRACE=1 bash tools/profile.sh r06_conc1 --concurrency 1 --steps 64 $Q > gpurun_out/r06_conc1.log 2>&1
bash tools/pmc.sh r06_pmc_prove --concurrency 1 --steps 6 $Q > gpurun_out/r06_pmc_prove.log 2>&1
bash tools/pmc.sh r06_pmc_prove23 --log2-size 23 --concurrency 1 --steps 4 $Q > gpurun_out/r06_pmc_prove23.log 2>&1
bash tools/pmc.sh r06_pmc_commit --workload commit --log2-size 26 --steps 12 --warmup 2 > gpurun_out/r06_pmc_commit.log 2>&1
PMC_COUNTERS=VALUBusy bash tools/pmc_valu.sh r06_valu --concurrency 1 --steps 6 $Q > gpurun_out/r06_valu.log 2>&1
python tools/kernel_breakdown.py "$(find gpurun_out/r06_conc1 -name '*kernel_trace.csv' | head -1)" 74 > gpurun_out/r06_kernel_breakdown.txt 2>&1
find gpurun_out/r06_conc1 -name '*kernel_trace.csv' -size +4M -delete
python tools/pmc_profile.py gpurun_out/r06_pmc_prove r06 prove 21 > gpurun_out/r06_post.log 2>&1
python tools/pmc_profile.py gpurun_out/r06_pmc_prove23 r06 prove 23 >> gpurun_out/r06_post.log 2>&1
python tools/pmc_profile.py gpurun_out/r06_pmc_commit r06 commit 26 14 >> gpurun_out/r06_post.log 2>&1
python tools/pmc_valu_profile.py gpurun_out/r06_valu r06 16 "bench.py prove m=21 --concurrency 1 --steps 6 (2 warm-up + 6 timed + 8 isolated proofs)" >> gpurun_out/r06_post.log 2>&1
mkdir -p gpurun_out/r06_out; cp profiles/r06_*pmc* gpurun_out/r06_out/ 2>/dev/null
find gpurun_out -name '*.csv' -size +3M -delete
sha256sum provekit_amd/lib/libprovekit_hip.so | cut -c1-16 > gpurun_out/r06_lib_sha16.txt
ls gpurun_out | head -40
