# same-box A/B: the committed tree (variants/head_tree, a git worktree of HEAD with its own build) against the working tree
for i in 1 2 3; do
  (cd variants/head_tree && python bench.py --no-cpu-baseline --camera-path 0 --steps 50 2>/dev/null) | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['stages_ms']; print('HEAD   ', d['ms_per_step'], d['step_ms'], 'reduce', s['reduce_partials'], 'bwd', s['blend_backward'], 'scan', s['scan_block_sums'])"
  python bench.py --no-cpu-baseline --camera-path 0 --steps 50 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['stages_ms']; print('WORKING', d['ms_per_step'], d['step_ms'], 'reduce', s['reduce_partials'], 'bwd', s['blend_backward'], 'scan', s['scan_block_sums'])"
done
