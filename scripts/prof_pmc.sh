#!/bin/bash
# One rocprofv3 --pmc pass per counter group (never combined with tracing), default bench workload.
# usage: scripts/prof_pmc.sh <tag> <group-name> <counters...>
set -o pipefail
TAG=$1; GROUP=$2; shift 2
OUT=$PWD/gpurun_out/prof_$TAG/pmc_$GROUP
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT" -- \
    python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "pmc $GROUP rc=$?"
