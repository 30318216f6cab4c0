#!/usr/bin/env python3
"""Per-launch kernel durations from a rocprofv3 --kernel-trace CSV directory.
usage: trace_kernels.py <dir> [name-substring ...]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
pats = sys.argv[2:] or ['k_']
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if any(p in n for p in pats):
        print('%-34s grid %-10s %.3f ms' % (n[:34], r.get('Grid_Size_X', ''), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
