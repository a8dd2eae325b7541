#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs produced by tools/pmc_run.sh: per kernel name, mean counter value per dispatch.
Kernels of this library (zlz4::*) and the runtime's buffer-fill kernel (hipMemsetAsync: the HC pipeline zeroes its result
area with it, and that traffic belongs to the compress call)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "zlz4" not in k and "fillBuffer" not in k:
            continue
        k = k.split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s n=%-3d mean %.6g" % (c, len(v), sum(v) / len(v)))
