#!/usr/bin/env python
"""GPU timeline of bench steps from a rocprofv3 kernel trace: per step, kernel count, busy time, span, and the
distribution of the gaps between consecutive kernels (is a small frame bound by its kernels, by the dependent-launch
boundaries, or by the host?).  usage: python tools/timeline_gaps.py <..._kernel_trace.csv>"""
import re
import sys

import pandas as pd

d = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp").reset_index(drop=True)
d["name"] = d.Kernel_Name.map(lambda k: (re.search(r"::(\w+)", k) or re.search(r"(\w+)", k)).group(1))
d["dur"] = (d.End_Timestamp - d.Start_Timestamp) / 1e3
d["gap"] = (d.Start_Timestamp - d.End_Timestamp.shift(1)) / 1e3
idx = d.index[d.name == "pose_inverse_kernel"].tolist()     # first kernel of an operator forward
steps = list(zip(idx[:-1], idx[1:]))[5:-1]                    # skip warm-up
rows = []
for a, b in steps:
    seg = d.iloc[a:b]
    rows.append(dict(kernels=len(seg), busy_us=seg.dur.sum(), step_us=(d.Start_Timestamp[b] - d.Start_Timestamp[a]) / 1e3,
                     gaps_us=seg.gap.iloc[1:].clip(lower=0).sum(), gap_median=seg.gap.iloc[1:].median(),
                     gap_max=seg.gap.iloc[1:].max()))
t = pd.DataFrame(rows)
print(t.describe().loc[["mean", "min", "max"]].round(1).to_string())
a, b = steps[len(steps) // 2]
print(d.iloc[a:b][["name", "dur", "gap"]].round(1).to_string())
