for r in 1 2; do for sp in 0 1; do
GS_FWD_SPLIT=$sp python bench.py --workload cfg1_10k_256 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg1 default-args GS_FWD_SPLIT=$sp ms %.4f strict %s steps %d median %.4f variants %s' % (d['ms_per_step'], d.get('ms_per_step_strict_warmup'), d['steps'], d['step_ms']['median'], {k:v['ms_per_step'] for k,v in d.get('variants',{}).items()}))"
GS_FWD_SPLIT=$sp python bench.py --workload cfg1_10k_256 --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg1 steps200 GS_FWD_SPLIT=$sp ms %.4f strict %s median %.4f' % (d['ms_per_step'], d.get('ms_per_step_strict_warmup'), d['step_ms']['median']))"
done; done
