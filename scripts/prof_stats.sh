#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench workload (run on the GPU box via gpurun).
# usage: scripts/prof_stats.sh <tag>
set -o pipefail
TAG=${1:-r1}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- \
    python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "stats rc=$?"
find "$OUT/stats" -name '*kernel_stats.csv' -exec cat {} \; | head -20
