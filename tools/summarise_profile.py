#!/usr/bin/env python
"""Condenses a gpurun_out/prof_<tag>/ directory (tools/profile.sh) into the small, committed summaries
under profiles/: per-kernel time statistics from the --kernel-trace --stats pass and per-kernel HBM
traffic from the two --pmc passes (FETCH_SIZE, WRITE_SIZE collected separately, as the MI355X guide
prescribes).  usage: python tools/summarise_profile.py <tag> [round-name]"""
import json
import os
import re
import sys

import pandas as pd

tag = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else tag
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def short(k):
    m = re.search(r"::(\w+)", k)
    return m.group(1) if m else k.split("(")[0][:60]


stats = pd.read_csv(os.path.join(src, "trace", "trace_kernel_stats.csv"))
stats["Name"] = stats["Name"].map(short)
stats.to_csv(os.path.join(dst, f"{name}_kernel_stats.csv"), index=False)

rows = {}
for pas, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    d = pd.read_csv(os.path.join(src, pas, f"{pas}_counter_collection.csv"))
    d = d[d.Counter_Name == ctr].copy()
    d["k"] = d.Kernel_Name.map(short)
    d["dur_us"] = (d.End_Timestamp - d.Start_Timestamp) / 1e3
    g = d.groupby("k").agg(launches=("Counter_Value", "count"), mean_kb=("Counter_Value", "mean"),
                           mean_us=("dur_us", "mean"))
    for k, r in g.iterrows():
        rows.setdefault(k, {})[ctr] = r.mean_kb
        rows[k][f"{pas}_pass_mean_us"] = r.mean_us
        rows[k]["launches"] = int(r.launches)
t = pd.DataFrame(rows).T
# rocprofv3 reports FETCH_SIZE/WRITE_SIZE in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section):
# FETCH_SIZE counts 128-B read requests as 64 B for wide coalesced streams -> reads are doubled; WRITE_SIZE
# is used as reported (uncalibrated).
t["hbm_read_MB_raw"] = t["FETCH_SIZE"] * 1024 / 1e6
t["hbm_read_MB_gfx950_corrected"] = 2 * t["hbm_read_MB_raw"]
t["hbm_write_MB"] = t["WRITE_SIZE"] * 1024 / 1e6
t = t.sort_values("fetch_pass_mean_us", ascending=False)
t.round(3).to_csv(os.path.join(dst, f"{name}_hbm_traffic.csv"))
# SQ counters (tools/profile.sh fourth pass): mean per launch and kernel
sq_path = os.path.join(src, "sq", "sq_counter_collection.csv")
if os.path.exists(sq_path):
    d = pd.read_csv(sq_path)
    d["kernel"] = d.Kernel_Name.map(short)
    sq = d.pivot_table(index="kernel", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
    sq["launches"] = d[d.Counter_Name == "SQ_WAVES"].groupby("kernel").size()
    sq.round(1).to_csv(os.path.join(dst, f"{name}_sq_counters.csv"))
    print(sq.round(0).to_string())
bench = {}
for pas in ("trace", "fetch", "write", "sq"):
    log = open(os.path.join(src, f"{pas}.log")).read()
    m = re.search(r'^\{"metric".*$', log, flags=re.M)
    if m:
        bench[pas] = json.loads(m.group(0))
json.dump(bench, open(os.path.join(dst, f"{name}_bench_lines.json"), "w"), indent=1)
# the kernel sources the profile was taken with (bench.py only quotes `traffic` / `valu` while they are the tree's)
hashes = {b.get("roofline", {}).get("kernel_source_hash") for b in bench.values()}
if len(hashes) == 1 and None not in hashes:
    with open(os.path.join(dst, f"{name}_source_hash.txt"), "w") as fh:
        fh.write(hashes.pop() + "  # sha256[:16] of csrc/*.hip,*.h at profile time (bench.py kernel_source_hash)\n")
pd.set_option("display.width", 200)
print(stats.head(14).to_string())
print(t.round(2).head(14).to_string())
