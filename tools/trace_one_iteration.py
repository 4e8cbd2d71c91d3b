#!/usr/bin/env python
"""Prints the kernel sequence of ONE steady-state training iteration from a rocprofv3 kernel trace CSV (development tool).
usage: python tools/trace_one_iteration.py <kernel_trace.csv> [anchor kernel substring] [which occurrence]"""
import sys

import pandas as pd

d = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp").reset_index(drop=True)
anchor = sys.argv[2] if len(sys.argv) > 2 else "controller_accumulate_kernel"
k = int(sys.argv[3]) if len(sys.argv) > 3 else 250
idx = d.index[d.Kernel_Name.str.contains(anchor)].tolist()
a, b = idx[k], idx[k + 1]
seg = d.iloc[a + 1:b + 1].copy()
seg["name"] = seg.Kernel_Name.str.replace("(anonymous namespace)::", "").str.replace("void ", "").str.replace("at::native::", "").str.slice(0, 80)
seg["us"] = (seg.End_Timestamp - seg.Start_Timestamp) / 1e3
seg["gap"] = (seg.Start_Timestamp - seg.End_Timestamp.shift(1)) / 1e3
print(seg[["name", "us", "gap"]].round(1).to_string(index=False))
print("kernels", len(seg), "busy us", round(seg.us.sum(), 1), "span us", round((seg.End_Timestamp.max() - d.End_Timestamp[a]) / 1e3, 1))
