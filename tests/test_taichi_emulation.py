"""The Taichi emulation the golden vectors are made with (tests/golden/taichi_emulation.py) is test infrastructure, but the
pins of the oracle stand on it: its execution model is checked here on a toy kernel.  A kernel's top-level loops are
separate offloaded tasks run in source order (the block loop on 256 threads per block with a real barrier and shared
memory, the loop behind it once, after the last block), atomics accumulate, and the results do not depend on how many
worker processes the blocks / iterations are dealt to (GS_EMU_PROCS)."""
import os
import sys
import textwrap

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import taichi_emulation as E  # noqa: E402

TOY = textwrap.dedent('''
    import taichi as ti


    @ti.kernel
    def two_loops(a: ti.types.ndarray(ti.f32, ndim=1), b: ti.types.ndarray(ti.f32, ndim=1),
                  total: ti.types.ndarray(ti.f32, ndim=1)):
        ti.loop_config(block_dim=256)
        for i in ti.ndrange(a.shape[0]):
            staged = ti.simt.block.SharedArray((256,), dtype=ti.f32)
            staged[i % 256] = a[i]
            ti.simt.block.sync()
            b[i] = staged[255 - i % 256]          # needs every thread of the block to have staged its value
            ti.atomic_add(total[0], 1.0)
        for j in range(b.shape[0]):               # a second offload: runs ONCE per j, after every block
            b[j] = b[j] + total[0]
''')


@pytest.mark.parametrize("procs", [1, 3])
def test_offloads_barriers_atomics_and_worker_processes(tmp_path, monkeypatch, procs):
    pkg = tmp_path / "taichi_3d_gaussian_splatting"
    pkg.mkdir()
    (pkg / "Toy.py").write_text(TOY)
    monkeypatch.setenv("GS_EMU_PROCS", str(procs))
    saved = {k: sys.modules.get(k) for k in ("taichi", "taichi.math", "dataclass_wizard", "taichi_3d_gaussian_splatting")}
    try:
        toy = E.load_reference(str(tmp_path), modules=("Toy",))["Toy"]
        n = 4096                                   # 16 blocks; long enough for the second loop to be dealt to the workers
        a = torch.arange(n, dtype=torch.float32)
        b, total = torch.zeros(n), torch.zeros(1)
        toy.two_loops(a, b, total)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    assert float(total[0]) == n
    i = np.arange(n)
    expected = (i - i % 256 + 255 - i % 256).astype(np.float32) + n
    assert np.array_equal(b.numpy(), expected)
