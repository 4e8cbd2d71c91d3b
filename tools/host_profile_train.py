#!/usr/bin/env python
"""Where the HOST time of a training iteration goes (the loop is host-bound at small frame sizes): wall-clock timers
around the Python entry points that run in the main thread and in the autograd thread (cProfile does not see the latter).
usage: python tools/host_profile_train.py [iterations=1500] [size=800]   (through gpurun)"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ITERS = sys.argv[1] if len(sys.argv) > 1 else "1500"
SIZE = sys.argv[2] if len(sys.argv) > 2 else "800"

acc = collections.defaultdict(lambda: [0.0, 0])
each = collections.defaultdict(list)
SHOW_EACH = {"controller._find_densify_points", "trainer._plot_grad_histogram", "trainer._plot_value_histogram",
             "controller._add_densify_points", "trainer.validation"}


def timed(owner, name, label=None):
    fn = getattr(owner, name)
    label = label or f"{getattr(owner, '__name__', type(owner).__name__)}.{name}"

    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = acc[label]
            dt = time.perf_counter() - t0
            e[0] += dt
            e[1] += 1
            if label in SHOW_EACH:
                each[label].append(round(dt * 1e3, 2))
    if isinstance(owner, type) and isinstance(owner.__dict__.get(name), staticmethod):
        wrapper = staticmethod(wrapper)
    setattr(owner, name, wrapper)


import torch  # noqa: E402
from taichi_3d_gaussian_splatting_amd import hip_ops, LossFunction as LF, optim  # noqa: E402
from taichi_3d_gaussian_splatting_amd import GaussianPointAdaptiveController as ADC  # noqa: E402
from taichi_3d_gaussian_splatting_amd import GaussianPointTrainer as TRNM  # noqa: E402

for name in ("pose_inverse", "filter_compact", "preprocess", "scan_block_sums_async", "make_keys", "sort_pairs",
             "tile_ranges", "blend_forward", "blend_backward", "point_backward", "controller_accumulate"):
    if hasattr(hip_ops, name):
        timed(hip_ops, name)
timed(hip_ops.CounterReadback, "wait", "CounterReadback.wait")
timed(ADC.GaussianPointAdaptiveController, "update", "controller.update (hook)")
timed(ADC.GaussianPointAdaptiveController, "refinement", "controller.refinement")
timed(ADC.GaussianPointAdaptiveController, "_find_densify_points", "controller._find_densify_points")
timed(ADC.GaussianPointAdaptiveController, "_add_densify_points", "controller._add_densify_points")
timed(TRNM.GaussianPointCloudTrainer, "_plot_grad_histogram", "trainer._plot_grad_histogram")
timed(TRNM.GaussianPointCloudTrainer, "_plot_value_histogram", "trainer._plot_value_histogram")
timed(TRNM.GaussianPointCloudTrainer, "validation", "trainer.validation")
timed(optim.Adam, "step", "Adam.step")
timed(TRNM.GaussianPointCloudTrainer, "_rasterise", "trainer._rasterise")
timed(torch.Tensor, "backward", "loss.backward (main thread, total)")
timed(LF.LossFunction, "forward", "LossFunction.forward")
for cls_name in dir(LF):
    cls = getattr(LF, cls_name)
    if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function:
        for m in ("forward", "backward"):
            f = cls.__dict__.get(m)
            if f is not None:
                inner = f.__func__ if isinstance(f, staticmethod) else f
                label = f"{cls_name}.{m}"

                def make(inner=inner, label=label):
                    def w(*a, **k):
                        t0 = time.perf_counter()
                        try:
                            return inner(*a, **k)
                        finally:
                            e = acc[label]; e[0] += time.perf_counter() - t0; e[1] += 1
                    return staticmethod(w)
                setattr(cls, m, make())

# steady state: the timers restart after the first validation (iteration 1000), once every kernel has been used
_validation = TRNM.GaussianPointCloudTrainer.validation
_state = {"reset_at": None}


def _validation_then_reset(self, loader, iteration):
    out = _validation(self, loader, iteration)
    if _state["reset_at"] is None:
        torch.cuda.synchronize()
        acc.clear(); each.clear()
        _state["reset_at"] = (iteration, time.perf_counter())
    return out


TRNM.GaussianPointCloudTrainer.validation = _validation_then_reset
sys.argv = ["train_7k.py", ITERS, "0", SIZE]
t0 = time.perf_counter()
exec(compile(open(os.path.join(ROOT, "tools", "train_7k.py")).read(), "train_7k.py", "exec"))
torch.cuda.synchronize()
n = int(ITERS)
if _state["reset_at"] is not None:
    n = int(ITERS) - _state["reset_at"][0]
    print(f"steady state: {n} iterations in {time.perf_counter() - _state['reset_at'][1]:.3f} s "
          f"= {(time.perf_counter() - _state['reset_at'][1]) / n * 1e6:.0f} us per iteration (incl. later validations)")
for label, v in each.items():
    print(f"{label} per call (ms): {v}")
print(f"\nhost timers over {n} iterations (us per iteration; calls per iteration)")
for label, (sec, calls) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"  {label:44s} {sec / n * 1e6:9.1f} us   {calls / n:5.2f} calls")
