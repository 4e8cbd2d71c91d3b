#!/bin/bash
# round 4, GPU call 18: kernel trace of the bench command with the MSD-first sort
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run18; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stage-profile > $OUT/trace.log 2>&1
find $OUT -name "*.db" -delete 2>/dev/null
f=$(find $OUT -name "*kernel_stats.csv" | head -1); cut -c1-100,200-400 $f | head -5; python - <<PY
import csv,re
rows=list(csv.DictReader(open("$f")))
for r in rows[:24]:
    m=re.search(r'(\w+_kernel)',r['Name']); print((m.group(1) if m else r['Name'][:40]).ljust(34), r['Calls'].rjust(5), "%9.1f us"%(float(r['AverageNs'])/1000), r['Name'][-60:] if 'sort' in r['Name'] else '')
PY
