#!/bin/bash
# (1) one rank, RCCL process group forced: the N > 1 code path (collective on the launch stream) on real hardware
# (2) two ranks sharing the one GPU of the box, gloo on DEVICE tensors (the path that hung in round 1), traced
set -o pipefail
mkdir -p gpurun_out/r2
RAY_BENCH_FORCE_PG=1 RAY_BENCH_TRACE=1 timeout -k 10 150 python bench.py --steps 10 --warmup 2 --no-cpu-baseline \
    > gpurun_out/r2/bench_pg1.json 2> gpurun_out/r2/bench_pg1.err
rc=$?; tail -c 600 gpurun_out/r2/bench_pg1.json; grep "rank" gpurun_out/r2/bench_pg1.err | tail -5; [ $rc -eq 0 ] || exit $rc
export RAY_BENCH_SHARE_GPU=1 RAY_BENCH_BACKEND=gloo RAY_BENCH_TRACE=1 RAY_BENCH_TRACE_DUMP_S=40
timeout -k 10 80 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/r2/bench_2r.json 2> gpurun_out/r2/bench_2r.err
rc=$?; tail -3 gpurun_out/r2/bench_2r.json; grep -v "^\[W\|amdgpu.ids" gpurun_out/r2/bench_2r.err | tail -60; exit $rc
