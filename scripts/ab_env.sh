#!/bin/bash
# A/B of environment knobs on the default bench workload: scripts/ab_env.sh "VAR=val" "VAR=val2" ...
for kv in "$@"; do
  env $kv timeout -k 10 120 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$kv', 'kernel_ms=%.3f'%d['roofline']['kernel_ms'], 'steps/s=%.3e'%d['roofline']['kernel_ray_steps_per_s'], 'frac=%.3f'%d['roofline']['frac'])" || echo "$kv FAILED"
done
