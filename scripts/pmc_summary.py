#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: per kernel, mean counter value per dispatch.
usage: scripts/pmc_summary.py <dir> [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_shoot"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if pat not in k:
            continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"  {c:40s} mean/dispatch {sum(v)/len(v):.6g}   (n={len(v)})")
