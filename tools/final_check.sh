cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/final/pytest.log 2>&1; tail -3 gpurun_out/final/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; python -c "
import json; d=json.load(open('gpurun_out/final/bench.json')); r=d['roofline']; print(d['ms_per_step'], d['value'], r['kernel'], r['frac'], 'traffic', r['traffic'], 'valu', r['valu'] and r['valu']['frac'], r['hbm_stage_furthest_from_bound']); print(d['cpu_baseline'])"
