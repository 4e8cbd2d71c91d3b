#!/usr/bin/env python
"""Registers, scratch and LDS of every kernel of one csrc/*.hip file as the compiler allocates them for gfx950 (development tool,
runs in the build container: hipcc --save-temps, then the .amdhsa metadata of the assembly).
usage: tools/kernel_regs.py gs_blend [extra hipcc flags ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, extra = sys.argv[1], sys.argv[2:]
src = os.path.join(ROOT, "taichi_3d_gaussian_splatting_amd", "csrc", name + ".hip")
per_file = ["-fno-slp-vectorize"] if name in ("gs_frontend", "gs_point_backward") else []
with tempfile.TemporaryDirectory() as tmp:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-munsafe-fp-atomics",
                    "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), *per_file, *extra, "-c", src, "--save-temps",
                    "-o", "x.o"], cwd=tmp, check=True, stderr=subprocess.DEVNULL)
    asm = open(os.path.join(tmp, f"{name}-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    keep = os.environ.get("KEEP_ASM")
    if keep:
        open(keep, "w").write(asm)
rows = []
for block in asm.split("  - .agpr_count:")[1:]:
    get = lambda k: re.search(r"\." + k + r":\s+(\S+)", block).group(1)   # noqa: E731
    sym = get("name")
    dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0]
    rows.append((dem, int(get("vgpr_count")), int(get("sgpr_count")), int(get("private_segment_fixed_size")),
                 int(get("group_segment_fixed_size"))))
for r in sorted(rows):
    print("%-70s vgpr %3d  sgpr %3d  scratch %3d  lds %6d" % r)
