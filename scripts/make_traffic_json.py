#!/usr/bin/env python3
"""profiles/<tag>/traffic.json from a scripts/prof_all.sh run: HBM bytes per k_shoot launch (FETCH_SIZE doubled per
MI355X_MICROARCH.md's gfx950 correction, WRITE_SIZE as is; both in KiB -> bytes), hit rates, VALU busy, and the
kernel's average duration from the kernel-trace stats of the same run.  usage: make_traffic_json.py <tag> <round>"""
import csv
import glob
import json
import os
import sys

tag, rnd = sys.argv[1], int(sys.argv[2])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
vals = {}
cur = None
for line in open(os.path.join(src, "pmc_summary.txt")):
    if not line.startswith(" "):
        cur = line.strip()
    elif cur and "k_shoot<1, 0, 1, 0>" in cur:
        name, _, rest = line.strip().partition(" ")
        vals[name] = float(rest.split("mean/dispatch")[1].split()[0])
avg_ms = None
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_shoot<1, 0, 1, 0>" in row["Name"]:
            avg_ms = float(row["AverageNs"]) * 1e-6
            calls, mn, mx = int(row["Calls"]), float(row["MinNs"]) * 1e-6, float(row["MaxNs"]) * 1e-6
fetch = vals["FETCH_SIZE"] * 1024 * 2
write = vals["WRITE_SIZE"] * 1024
xcd_cycles = vals["GRBM_GUI_ACTIVE"] / 8
out = {
    "round": rnd, "kernel": "k_shoot<grid,none,fused>",
    "source": f"profiles/{tag}/pmc_summary.txt + kernel_stats.csv (rocprofv3 --pmc, one pass per counter group, and "
              "--kernel-trace --stats: scripts/prof_all.sh; bench.py --steps 3 for the counters, --steps 40 --warmup 5 for the kernel trace)",
    "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
    "fetch_bytes_corrected": int(fetch), "write_bytes": int(write), "traffic_bytes_per_launch": int(fetch + write),
    "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B): an upper estimate for "
            "the gather-shaped table reads.  Algorithmic HBM bytes per launch: 80 B/ray (9 f64 + 2 int32) x 4194304 "
            "rays = 335.5 MB, + 32 B/ray parked pre-exit state written and read back once.",
    "l2_hit_rate": round(vals["TCC_HIT_sum"] / vals["TCC_REQ_sum"], 4),
    "l1_hit_rate": round(1 - vals["TCP_TCC_READ_REQ_sum"] / vals["TCP_TOTAL_CACHE_ACCESSES_sum"], 4),
    "valu_insts_per_launch": vals["SQ_INSTS_VALU"], "lds_insts_per_launch": vals["SQ_INSTS_LDS"],
    "vmem_insts_per_launch": vals["SQ_INSTS_VMEM"], "salu_insts_per_launch": vals["SQ_INSTS_SALU"],
    "valu_busy_frac": round(vals["SQ_ACTIVE_INST_VALU"] * 4 / (xcd_cycles * 1024), 4),
    "valu_busy_formula": "SQ_ACTIVE_INST_VALU * 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)",
    "lds_bank_conflict_cycles": vals["SQ_LDS_BANK_CONFLICT"], "lds_active_cycles": vals["SQ_ACTIVE_INST_LDS"],
    "wait_inst_lds_frac_of_wave_cycles": round(vals["SQ_WAIT_INST_LDS"] / vals["SQ_WAVE_CYCLES"], 4),
    "wait_inst_any_frac_of_wave_cycles": round(vals["SQ_WAIT_INST_ANY"] / vals["SQ_WAVE_CYCLES"], 4),
    "mean_waves_per_simd": round(vals["SQ_WAVE_CYCLES"] * 4 / (xcd_cycles * 1024), 3),
    "kernel_avg_ms": avg_ms, "kernel_min_ms": mn, "kernel_max_ms": mx, "kernel_calls": calls,
    "waves": vals["SQ_WAVES"],
}
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
json.dump(out, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    open(os.path.join(dst, "kernel_stats.csv"), "w").write(open(f).read())
for f in glob.glob(os.path.join(src, "stats_secondary", "**", "*kernel_stats.csv"), recursive=True):
    open(os.path.join(dst, "kernel_stats_secondary.csv"), "w").write(open(f).read())
for a, b in (("bench.json", "bench_under_rocprof.json"), ("bench_secondary.json", "bench_secondary_under_rocprof.json")):
    if os.path.exists(os.path.join(src, a)):
        open(os.path.join(dst, b), "w").write(open(os.path.join(src, a)).read())
open(os.path.join(dst, "pmc_summary.txt"), "w").write(open(os.path.join(src, "pmc_summary.txt")).read())
print(json.dumps(out, indent=1))
