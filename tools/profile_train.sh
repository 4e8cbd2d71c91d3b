#!/bin/bash
# rocprofv3 kernel statistics of the training loop (tools/train_7k.py, HIP back end only).  usage: tools/profile_train.sh [iterations=2000] [size=800]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_train
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $ROOT/tools/train_7k.py ${1:-2000} 0 ${2:-800} > $OUT/log.txt 2>&1
cd $ROOT
grep -E "hip_it_per_s|hip_seconds" $OUT/log.txt | head -3
python - <<PY
import pandas as pd, re
d = pd.read_csv("$OUT/t_kernel_stats.csv")
d["Name"] = d.Name.map(lambda k: (re.search(r"::(\w+)", k) or re.search(r"(\w+)", k)).group(1))
print(d.iloc[:32, :5].to_string())
print("total kernel ms", d.TotalDurationNs.sum() / 1e6, "calls", d.Calls.sum())
PY
rm -f $OUT/*trace.csv
