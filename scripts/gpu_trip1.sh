#!/bin/bash
# round-2 trip 1: parity suite + the WIP no-alloc test + a bench line
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/gpu_tests.log 2>&1
rc=$?; tail -5 gpurun_out/r2/gpu_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > gpurun_out/r2/bench1.json 2> gpurun_out/r2/bench1.err
rc=$?; cat gpurun_out/r2/bench1.json; [ $rc -eq 0 ] || { tail -5 gpurun_out/r2/bench1.err; exit $rc; }
