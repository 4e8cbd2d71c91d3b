#!/usr/bin/env python
"""Makes the trained 1920 x 1072 workload (trained_workload.py) and prints how it grew (development tool; gpurun)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taichi_3d_gaussian_splatting_amd import host_affinity  # noqa: E402
from taichi_3d_gaussian_splatting_amd.trained_workload import load_or_make  # noqa: E402

host_affinity.pin_host_threads(0)
made = load_or_make("trained_1080p", verbose=True)
print("[trained_scene]", json.dumps(made["stats"]), flush=True)
