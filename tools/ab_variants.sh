#!/bin/bash
# Same-box A/B of library variants (tools/build_variants.sh): alternating runs of the bench line, `rounds` times.
# usage: tools/ab_variants.sh <rounds> <tag> [<tag> ...]     (tag "product" = the in-tree library)
# env: AB_ARGS = extra bench.py arguments (default: the headline workload)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for tag in "$@"; do
    if [ "$tag" = product ]; then lib=""; else lib=$(pwd)/variants/libgsplat_hip_$tag.so; fi
    GS_LIB_PATH=$lib GS_ALLOW_TUNING_LIB=1 python bench.py --no-cpu-baseline --camera-path 0 --steps 50 $AB_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=(d.get('roofline') or {}).get('stages_ms') or {}
print('%-10s ms_per_step %.4f  median %.4f  fwd %.4f  bwd %.4f  reduce %.4f  pre %.4f  sort %.4f  keys %.4f  pointbwd %.4f' % ('$tag', d['ms_per_step'], d['step_ms']['median'], s.get('blend_forward',0), s.get('blend_backward',0), s.get('reduce_partials',0), s.get('preprocess',0), s.get('sort_pairs',0), s.get('make_keys',0), s.get('point_backward',0)))"
  done
done
