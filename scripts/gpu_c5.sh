#!/bin/bash
# BASELINE config 5's grid: the reconstruction loop on a 1024^2 grid, matrix-free, one rank's shard on one GPU
set -o pipefail
mkdir -p gpurun_out/r2
S=examples/distributed_reconstruction.py
timeout -k 10 900 python $S --ns ${C5_NS:-64} --nd ${C5_ND:-64} --N 1024 --iters ${C5_ITERS:-3} --solver cg --cg-rtol ${C5_RTOL:-1e-6} --cg-maxit ${C5_MAXIT:-400} --time-step ${C5_DT:-1e-2} > gpurun_out/r2/c5_1024.jsonl 2> gpurun_out/r2/c5_1024.err
rc=$?; cat gpurun_out/r2/c5_1024.jsonl; [ $rc -eq 0 ] || tail -8 gpurun_out/r2/c5_1024.err; exit $rc
