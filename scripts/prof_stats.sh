#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench workload (run on the GPU box via gpurun): the headline loop alone
# (stats/: every k_shoot<1,0,1,0> call is a headline launch, so its average is the figure to hold against bench.py's
# live kernel_ms) and the same command with the `secondary` kernels (stats_secondary/: config 3, analytic, dt = h/2,
# a9 trace, Frechet build, loop passes -- there k_shoot<1,0,1,0> also carries the dt = h/2 launches).
# usage: scripts/prof_stats.sh <tag>
set -o pipefail
TAG=${1:-r1}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- \
    python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "stats rc=$?"
find "$OUT/stats" -name '*kernel_stats.csv' -exec cat {} \; | head -6
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_secondary" -- \
    python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline > "$OUT/bench_secondary.json" 2> "$OUT/bench_secondary.err"
echo "stats (with secondary) rc=$?"
find "$OUT/stats_secondary" -name '*kernel_stats.csv' -exec cat {} \; | head -12
