#!/bin/bash
# A/B of library variants on the default bench workload: scripts/ab.sh exp_v0.so exp_v1.so ...
# (separate processes: quote only large differences; see cdna_hip_programming.md rule 24)
for lib in "$@"; do
  RAY_AMD_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', 'kernel_ms=%.3f'%d['roofline']['kernel_ms'], 'steps/s=%.3e'%d['roofline']['kernel_ray_steps_per_s'], 'frac=%.3f'%d['roofline']['frac'])" || echo "$lib FAILED"
done
