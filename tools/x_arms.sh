# backward reduce arms: pair DPP reduce-scatter (default) / per-entry DPP / per-entry through the matrix pipe
for v in "" single mfma; do
  if [ -z "$v" ]; then unset GS_LIB_PATH; else export GS_LIB_PATH=variants/libgsplat_hip_$v.so; fi
  python bench.py --no-cpu-baseline --camera-path 0 --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['stages_ms']; print('${v:-pair(default)}', 'ms_per_step', d['ms_per_step'], 'blend_backward_ms', s['blend_backward'])"
done
