#!/bin/bash
# SQ counters of the blend kernels for a library variant (same-box comparisons of arms): tools/pmc_blend.sh <outdir> <tag> [<tag> ...]
# One --pmc pass per variant, kernel trace only (never combined with other trace domains).
OUT=$(pwd)/$1; shift
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  if [ "$tag" = product ]; then lib=""; else lib=$ROOT/variants/libgsplat_hip_$tag.so; fi
  GS_LIB_PATH=$lib GS_ALLOW_TUNING_LIB=1 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES \
      --kernel-trace --output-format csv -d $OUT/pmc_$tag -o pmc -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --camera-path 0 --no-stage-profile > $OUT/pmc_$tag.log 2>&1
  python - <<PY
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "blend" in k:
            k = k.split("<")[0].split("::")[-1]
            rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in rows.items():
    print("$tag", k, {n: round(sum(v) / len(v) / 1e6, 2) for n, v in sorted(c.items())}, "launches", len(next(iter(c.values()))))
PY
  find $OUT -name "*.db" -delete 2>/dev/null
done
