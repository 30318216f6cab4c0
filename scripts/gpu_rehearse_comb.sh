#!/bin/bash
# two ranks time-slicing one GPU (gloo, host-side reduction): bench.py's N > 1 path with theta-comb sharding,
# reduced sums verified against the ranks' own inside bench.py
set -o pipefail
mkdir -p gpurun_out/r3
export RAY_BENCH_SHARE_GPU=1 RAY_BENCH_BACKEND=gloo
timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r3/bench_2r_comb.json 2> gpurun_out/r3/bench_2r_comb.err
rc=$?; cat gpurun_out/r3/bench_2r_comb.json | cut -c1-900; [ $rc -eq 0 ] || tail -20 gpurun_out/r3/bench_2r_comb.err; exit $rc
