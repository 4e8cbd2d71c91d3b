#!/usr/bin/env python
"""Fills the R6_* placeholders of DESIGN.md / README.md / BASELINE.md from profiles/r06_bench_all_configs.jsonl and
profiles/r06_trained_scene.txt (run once the round's report has been copied into profiles/).  usage: python tools/fill_round_numbers.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lines = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", "r06_bench_all_configs.jsonl"))]


def pick(workload, fwd_only=False, nth=0):
    want = "(fwd)" if fwd_only else "(fwd+bwd)"
    return [d for d in lines if d["config"].get("workload") == workload and want in d["metric"]][nth]


def both(d):
    return f"{d['ms_per_step']:.4f} ({d.get('ms_per_step_strict_warmup', float('nan')):.4f})"


head = pick("headline_1m_1080p", nth=0)
heads = [d for d in lines if d["config"].get("workload") == "headline_1m_1080p" and "(fwd+bwd)" in d["metric"]]
fps = re.search(r'"ms_per_frame": ([0-9.]+), "fps": ([0-9.]+), "points": (\d+)', open(os.path.join(ROOT, "profiles", "r06_trained_scene.txt")).read())
vals = {
    "R6_HEAD_LEAN": f"{head['variants']['hook_without_feature_copy']['ms_per_step']:.4f}",
    "R6_HEAD_STATIC": f"{heads[1]['ms_per_step']:.4f}", "R6_HEAD_NOHOOK": f"{heads[2]['ms_per_step']:.4f}",
    "R6_HEAD_MPX": f"{head['value']:.0f}", "R6_HEAD_CAM": f"{head['variants']['camera_path']['ms_per_step']:.4f}",
    "R6_HEAD": both(head),
    "R6_FWD_RGB": f"{pick('headline_1m_1080p', True, 1)['ms_per_step']:.3f}", "R6_FWD": f"{pick('headline_1m_1080p', True, 0)['ms_per_step']:.3f}",
    "R6_CFG1_MPX": f"{pick('cfg1_10k_256')['value']:.0f}", "R6_CFG1_CAM": f"{pick('cfg1_10k_256')['variants']['camera_path']['ms_per_step']:.3f}",
    "R6_CFG1": both(pick("cfg1_10k_256")), "R6_CFG2": f"{pick('cfg2_100k_800')['ms_per_step']:.3f}",
    "R6_CFG3": f"{pick('cfg3_400k_1080p')['ms_per_step']:.3f}", "R6_CFG4": f"{pick('cfg4_2m_1080p')['ms_per_step']:.3f}",
    "R6_STRESS_FWD": f"{pick('stress_t_ras', True)['ms_per_step']:.3f}", "R6_STRESS": f"{pick('stress_t_ras')['ms_per_step']:.3f}",
    "R6_TRAINED_FWD": f"{pick('trained_1080p', True)['ms_per_step']:.3f}",
    "R6_TRAINED_FPS": f"{float(fps.group(1)):.3f} ms per frame = {float(fps.group(2)):.0f} FPS" if fps else "n/a",
    "R6_TRAINED": f"{pick('trained_1080p')['ms_per_step']:.3f}",
}
for name in ("DESIGN.md", "README.md", "BASELINE.md"):
    path = os.path.join(ROOT, name)
    text = open(path).read()
    for k in sorted(vals, key=len, reverse=True):
        text = text.replace(k, vals[k])
    open(path, "w").write(text)
print(json.dumps(vals, indent=1))
