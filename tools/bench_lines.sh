#!/bin/bash
# The bench lines the documents quote (part 2 of tools/round_report.sh on its own).  usage: tools/bench_lines.sh <outdir>
OUT=${1:-gpurun_out/bench_lines}
mkdir -p $OUT
: > $OUT/bench_all_configs.jsonl
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cat $OUT/bench_default.json >> $OUT/bench_all_configs.jsonl
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2>> $OUT/bench.err
for w in cfg1_10k_256 cfg2_100k_800 cfg3_400k_1080p cfg4_2m_1080p stress_t_ras trained_1080p; do
    python bench.py --workload $w --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
done
python bench.py --static-scene --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
python bench.py --no-hook --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
python bench.py --no-hook-feature-copy --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
for w in headline_1m_1080p cfg3_400k_1080p stress_t_ras trained_1080p; do
    python bench.py --workload $w --forward-only --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
    python bench.py --workload $w --forward-only --rgb-only --no-cpu-baseline 2>> $OUT/bench.err >> $OUT/bench_all_configs.jsonl
done
wc -l $OUT/bench_all_configs.jsonl
