import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from taichi_3d_gaussian_splatting_amd import host_affinity
from taichi_3d_gaussian_splatting_amd.trained_workload import make_trained_scene
host_affinity.pin_host_threads(0)
for kw in (dict(n_true=300_000, init_fraction=0.1, densify_threshold=1e-6),
           dict(n_true=300_000, init_fraction=0.1, densify_threshold=3e-7),
           dict(n_true=150_000, init_fraction=0.2, densify_threshold=1e-7)):
    t0 = time.time()
    made = make_trained_scene(max_iterations=5001, **kw)
    st = made["stats"]
    print("[probe]", kw, "->", st["live_gaussians"], "iters", st["iterations"], "growth", st["growth"][::5], "aniso", round(st["anisotropy_median"], 2), round(st["anisotropy_p99"], 1), "opacity", round(st["opacity_median"], 3), "secs", round(time.time() - t0, 1), flush=True)
