#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output (counter_collection.csv [+ kernel_trace.csv]):
mean counter value and mean duration per kernel name / grid size."""
import collections
import csv
import glob
import re
import sys

path = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+Kernel(?:<[^>]*>)?)", r["Kernel_Name"])
        name = (m.group(1) if m else r["Kernel_Name"][:40]) + " grid=" + r.get("Grid_Size", "")
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r and r["Start_Timestamp"]:
            dur[name][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for name, c in agg.items():
    d = list(dur[name].values())
    print("%s  n=%d  avg %.1f us" % (name, len(d), sum(d) / max(len(d), 1) / 1e3))
    for k, v in sorted(c.items()):
        print("    %-30s %.5g" % (k, sum(v) / len(v)))
