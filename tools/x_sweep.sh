for v in "" w7 w7b5; do
  if [ -z "$v" ]; then unset GS_LIB_PATH; else export GS_LIB_PATH=variants/libgsplat_hip_$v.so; fi
  python bench.py --no-cpu-baseline --steps 30 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['stages_ms']; print('${v:-default}', d['ms_per_step'], 'pre', s['preprocess'], 'fwd', s['blend_forward'], 'bwd', s['blend_backward'])"
done
