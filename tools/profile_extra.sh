#!/bin/bash
# Kernel statistics (rocprofv3 --kernel-trace --stats) of the bench lines other than the default one: forward-only
# (inference), the stress distribution, cfg 3 and cfg 4.  usage: tools/profile_extra.sh <tag>  -> gpurun_out/prof_<tag>_extra/
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_extra
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {   # name, bench arguments
    name=$1; shift
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o t -- \
        python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-stage-profile "$@" > $OUT/$name.log 2>&1
    cp $OUT/$name/t_kernel_stats.csv $OUT/${name}_kernel_stats.csv 2>/dev/null
    rm -rf $OUT/$name
}
run forward_only --forward-only
run stress --workload stress_t_ras
run cfg3 --workload cfg3_400k_1080p
run cfg4 --workload cfg4_2m_1080p
ls -la $OUT
