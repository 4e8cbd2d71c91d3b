#!/usr/bin/env python
"""rocprofv3 kernel trace -> median duration per (kernel, number of workgroups).  usage: trace_by_grid.py <dir> [...]"""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    agg = collections.defaultdict(list)
    for path in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"]
            short = name.split("(")[0].split("::")[-1][:44] if "::" in name.split("(")[0] else name[:44]
            wg = max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))), 1)
            grid = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
            agg[(short, grid // wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("==", d)
    for k, v in sorted(agg.items()):
        v.sort()
        print(f"  {k[0]:46s} blocks={k[1]:6d} calls={len(v):4d} median={v[len(v) // 2]:8.1f}us min={v[0]:8.1f}us")
