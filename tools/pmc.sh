#!/bin/bash
# PMC counter pass over tools/stage_bench.py (development tool; run through gpurun).
# usage: tools/pmc.sh <tag> "<COUNTER LIST>"
set -u
TAG=$1; CTRS=$2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -o pmc -- python $ROOT/tools/stage_bench.py headline_1m_1080p 3 > $OUT/log.txt 2>&1
python - <<PY
import pandas as pd, glob
f = glob.glob("$OUT/**/pmc_counter_collection.csv", recursive=True)[0]
d = pd.read_csv(f)
d = d[d.Kernel_Name.str.contains("blend_|sort_scatter|preprocess|point_backward")]
d["k"] = d.Kernel_Name.str.extract(r"::(\w+)")
t = d.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
pd.set_option("display.width", 250); pd.set_option("display.float_format", lambda v: f"{v:,.0f}")
print(t.to_string())
PY
