#!/bin/bash
# Frechet pipeline profile: kernel trace of scripts/bench_frechet_large.py (512 x 512 rays, N = 513, dt = 1e-2)
# plus the loop passes; usage: prof_frechet.sh <tag>   -> gpurun_out/<tag>/
set -e
TAG=${1:-frechet}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python3 $R/scripts/bench_frechet_large.py 512 512 513 1e-2 > $R/$OUT/large.jsonl 2> $R/$OUT/large.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt -o kt -- python3 $R/scripts/bench_frechet_large.py 512 512 513 1e-2 > $R/$OUT/kt.log 2>&1
find $R/$OUT/kt -name '*kernel_stats.csv' -exec cp {} $R/$OUT/kernel_stats.csv \;
rm -rf $R/$OUT/kt
cat $R/$OUT/large.jsonl
head -12 $R/$OUT/kernel_stats.csv
