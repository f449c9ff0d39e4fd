#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace summary of the default bench command -> gpurun_out/<tag>/
# usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-prof}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
mkdir -p "$R/gpurun_out/$TAG"
cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d "$R/gpurun_out/$TAG" -o bench -- python "$R/bench.py" --no-cpu-baseline "$@" > "$R/gpurun_out/$TAG/bench.log" 2>&1
grep '^{' "$R/gpurun_out/$TAG/bench.log" > "$R/gpurun_out/$TAG/bench.json"
# keep only the summaries (the full kernel trace is large)
[ -n "${KEEP_TRACE:-}" ] || find "$R/gpurun_out/$TAG" -name '*kernel_trace.csv' -size +4M -delete
ls -la "$R/gpurun_out/$TAG"
head -25 "$R/gpurun_out/$TAG"/*kernel_stats.csv 2>/dev/null | cut -c1-220
