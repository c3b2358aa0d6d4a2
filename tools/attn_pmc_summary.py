#!/usr/bin/env python3
"""Reduce the rocprofv3 counter CSVs of tools/attn_pmc.sh / tools/mid_pmc.sh to one table: per-launch average of every counter for one
kernel (argv[2]: a substring of its name, default attn_resident_k)."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "attn_resident_k"
acc, cnt = defaultdict(float), defaultdict(int)
for f in sorted(glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if KERNEL not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[r["Counter_Name"]] += 1
# one CSV row per (dispatch, counter[, dimension]) — sum the dimensions of a dispatch, average over dispatches
disp = defaultdict(set)
for f in sorted(glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            disp[r["Counter_Name"]].add((f, r["Dispatch_Id"]))
print(f"{'counter':28s} {'per launch':>16s}  launches")
for k in sorted(acc):
    n = max(1, len(disp[k]))
    print(f"{k:28s} {acc[k] / n:16.0f}  {n}")
for f in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Name"]:
            print(f"kernel-trace: calls {r['Calls']} average {float(r['AverageNs']) / 1e3:.1f} us (min {float(r['MinNs']) / 1e3:.1f}, max {float(r['MaxNs']) / 1e3:.1f})")
w = acc.get("SQ_WAVE_CYCLES", 0)
if w:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY"):
        print(f"{k} / SQ_WAVE_CYCLES = {acc[k] / w:.3f}")
    if acc.get("SQ_LDS_IDX_ACTIVE"):
        print(f"SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = {acc['SQ_LDS_BANK_CONFLICT'] / acc['SQ_LDS_IDX_ACTIVE']:.3f}")
