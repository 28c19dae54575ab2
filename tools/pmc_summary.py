#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (tools/prof_pmc.sh) per kernel: mean of every
counter over the dispatches of each kernel.  FETCH_SIZE / WRITE_SIZE are
reported raw (KiB per the rocprofv3 definition) and in bytes with the gfx950
correction of MI355X_MICROARCH.md (FETCH_SIZE x2 for wide coalesced reads is
NOT applied blindly: both figures are printed).
    python3 tools/pmc_summary.py OUTDIR > profiles/rNN_pmc_summary.csv"""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
# optional 2nd argument: a note (frame size, build) written as a leading comment line
if len(sys.argv) > 2:
    print("# " + sys.argv[2])
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        rd = csv.DictReader(fh)
        per = defaultdict(float)
        for row in rd:
            k = (row["Dispatch_Id"], row["Kernel_Name"], row["Counter_Name"])
            per[k] += float(row["Counter_Value"])
        for (d, kern, ctr), v in per.items():
            acc[kern][ctr].append(v)
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
for kern in sorted(acc):
    for ctr in sorted(acc[kern]):
        v = acc[kern][ctr]
        w.writerow([kern, ctr, len(v), "%.6g" % (sum(v) / len(v))])
