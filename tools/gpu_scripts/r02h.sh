#!/bin/bash
# Round 2: every kernel of DESIGN section 5 re-measured in one lease (rocprofv3 kernel stats around tools/bench_configs.py, both
# layouts), the (9,3) A/B, the BASELINE configs[4] step end to end, the resampler shapes.
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r02h
R=$GRAFT_REPO_ROOT
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg -- python $R/tools/bench_configs.py --configs 3456789a --layouts soa,aos > $O/prof_cfg.log 2>&1; echo "cfg rc=$?"
cd $R
grep -E "^\{" $O/prof_cfg.log > $O/configs_all.jsonl; wc -l $O/configs_all.jsonl
timeout 300 python tools/exp_ml.py > $O/c3_multilane.jsonl 2> $O/c3_multilane.err; echo "ml rc=$?"
timeout 300 python tools/bench_c5.py > $O/bench_c5_1000x8000.json 2> $O/bench_c5_a.err; echo "c5a rc=$?"
timeout 300 python tools/bench_c5.py --filters 125 --particles 8000 > $O/bench_c5_125x8000.json 2> $O/bench_c5_b.err; echo "c5b rc=$?"
timeout 300 python tools/bench_c5.py --filters 125 --particles 8000000 --steps 5 --warmup 2 > $O/bench_c5_125x8e6.json 2> $O/bench_c5_c.err; echo "c5c rc=$?"
timeout 300 python tools/bench_resample.py --shapes 125x8000000,8x8000000,1x8000000,1000x8000,125x8000 --iters 10 > $O/resample_shapes.jsonl 2>/dev/null
find $O -name "*kernel_trace.csv" -size +1M -delete
cat $O/configs_all.jsonl | cut -c1-200 | head -80
cat $O/bench_c5_*.json | cut -c1-400
