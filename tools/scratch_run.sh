for r in 1 2 3; do for sk in 1 0; do
GS_BWD_FORM_BY_SKEW=$sk python bench.py --no-cpu-baseline --camera-path 0 --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline by_skew=$sk ms_per_step %.4f strict %s median %.4f p90 %.4f' % (d['ms_per_step'], d.get('ms_per_step_strict_warmup'), d['step_ms']['median'], d['step_ms']['p90']))"
done; done
