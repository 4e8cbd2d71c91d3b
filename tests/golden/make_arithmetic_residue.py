#!/usr/bin/env python
"""How much of the REFERENCE's own output depends on arithmetic liberties its real back end may take and this repository's
IEEE emulation does not.

Real Taichi compiles the reference's kernels with ``fast_math=True`` (no ``ti.init`` in /root/reference turns it off) and a
device ``exp``: FMA contraction, division by reciprocal-multiply and a non-correctly-rounded exponential are all allowed to
it.  The committed reference-run vectors (``reference_operator_*_exp_cr.npz``) are the reference's unmodified sources under
IEEE fp32 with a correctly rounded exp.  This script re-runs the same sources on the same scenes with ONE liberty taken
everywhere it syntactically can be (taichi_emulation.py: GS_EMU_FMA, GS_EMU_RCP_DIV; GS_EMU_EXP unset = NumPy's fp32 exp)
and reports, against the committed IEEE run: pixels whose number of blended Gaussians changes (a discrete decision moved:
alpha >= 1/255, RAS:451, or T' < 1e-4, RAS:458), Gaussians whose affected-pixel count changes (RAS:631), image L-inf, and
relative L2 of the gradients.  That is the yardstick for "zero flipped decisions against the IEEE run": what the reference
itself moves by when its own arithmetic is varied within what its compiler may do.

Build container only (executes /root/reference; each (scene, arithmetic) run takes minutes):
    GS_EMU_PROCS=8 python tests/golden/make_arithmetic_residue.py [scene name prefixes, default j k p]
-> tests/golden/arithmetic_residue.json (+ the table printed for tests/golden/README.md)
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ARITHMETICS = {
    "fma_contraction": {"GS_EMU_FMA": "1", "GS_EMU_EXP": "cr"},
    "division_by_reciprocal": {"GS_EMU_RCP_DIV": "1", "GS_EMU_EXP": "cr"},
    "fma_and_reciprocal": {"GS_EMU_FMA": "1", "GS_EMU_RCP_DIV": "1", "GS_EMU_EXP": "cr"},
    "numpy_fp32_exp": {"GS_EMU_EXP": ""},
}


def compare(base, other):
    rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))   # noqa: E731
    same_visible = np.array_equal(base["hook_point_id"], other["hook_point_id"])
    out = {
        "pixels": int(base["count"].size),
        "pixels_with_another_count": int((base["count"] != other["count"]).sum()),
        "image_linf": float(np.abs(base["image"] - other["image"]).max()),
        "image_linf_on_pixels_with_the_same_count": float(
            (np.abs(base["image"] - other["image"]).max(axis=2) * (base["count"] == other["count"])).max()),
        "depth_linf": float(np.abs(base["depth"] - other["depth"]).max()),
        "same_visible_set": bool(same_visible),
        "grad_xyz_rel_l2": rel(other["grad_xyz"], base["grad_xyz"]),
        "grad_feat_rel_l2": rel(other["grad_feat"], base["grad_feat"]),
    }
    if same_visible:
        out["gaussians"] = int(base["hook_point_id"].size)
        out["gaussians_with_another_tile_count"] = int((base["hook_num_overlap_tiles"] != other["hook_num_overlap_tiles"]).sum())
        out["gaussians_with_another_affected_pixel_count"] = int(
            (base["hook_num_affected_pixels"] != other["hook_num_affected_pixels"]).sum())
    return out


def main():
    sys.path.insert(0, HERE)
    from make_reference_operator_vectors import SCENES
    wanted = sys.argv[1:] or ["j_", "k_", "p_"]
    names = [n for n in SCENES if any(n.startswith(w) for w in wanted)]
    path = os.path.join(HERE, "arithmetic_residue.json")
    table = json.load(open(path)) if os.path.exists(path) else {}
    for name in names:
        base = np.load(os.path.join(HERE, f"reference_operator_{name}_exp_cr.npz"))
        for tag, env in ARITHMETICS.items():
            if tag in table.get(name, {}):
                continue
            with tempfile.TemporaryDirectory() as tmp:
                e = dict(os.environ, GS_EMU_OUT_DIR=tmp, **env)
                if not env.get("GS_EMU_EXP"):
                    e.pop("GS_EMU_EXP", None)
                    e["GS_EMU_EXP"] = "numpy"     # (anything but "cr": the generator only setdefaults it)
                subprocess.run([sys.executable, os.path.join(HERE, "make_reference_operator_vectors.py"), name], env=e, check=True,
                               stdout=subprocess.DEVNULL)
                produced = [f for f in os.listdir(tmp) if f.endswith(".npz")]
                assert len(produced) == 1, produced
                other = np.load(os.path.join(tmp, produced[0]))
                table.setdefault(name, {})[tag] = compare(base, other)
            print(name, tag, json.dumps(table[name][tag]), flush=True)
            with open(path, "w") as fh:
                json.dump(table, fh, indent=1, sort_keys=True)
    print("\n| scene | arithmetic | pixels with another blended set | Gaussians with another affected-pixel count | image L-inf "
          "(all / same-count pixels) | grad rel-L2 (xyz / features) |\n|---|---|---|---|---|---|")
    for name in table:
        for tag, r in table[name].items():
            print(f"| `{name.split('_')[0]}` | {tag} | {r['pixels_with_another_count']} of {r['pixels']} | "
                  f"{r.get('gaussians_with_another_affected_pixel_count', 'n/a')} of {r.get('gaussians', 'n/a')} | "
                  f"{r['image_linf']:.1e} / {r['image_linf_on_pixels_with_the_same_count']:.1e} | "
                  f"{r['grad_xyz_rel_l2']:.1e} / {r['grad_feat_rel_l2']:.1e} |")


if __name__ == "__main__":
    main()
