#!/usr/bin/env python3
"""Prints the kernels of a rocprofv3 --kernel-trace csv as a timeline (ms since the first): tools/timeline.py <kernel_trace.csv> [first] [count]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[first:first + count]:
    n = r["Kernel_Name"]
    n = n[:n.index("(")] if "(" in n else n
    print("q%-2s %10.3f %9.3f  grid %-7s %s" % (r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Grid_Size_X"], n[-70:]))
