#!/bin/bash
# Round 6, first lease: the whole GPU suite on the round's first library (custom inv, IMM banks of 9..16, one-pass v2), smoke,
# the bench line with its new "configs" rows, and the one-pass resampler A/B (FK_OP_V2=0 = round 3's kernel) with phase clocks.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_scripts/r06_a.sh'
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06a
mkdir -p $O
export TMPDIR=/tmp
cd $R
bash tools/gpu_scripts/box_state.sh > $O/box_state.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-400 $O/bench_default.json; tail -3 $O/bench_default.err
SH=125x8000000,8x8000000,1x8000000,1000x100000,32x1000000
for v in 0 1 0 1; do
  FK_OP_V2=$v timeout 200 python tools/bench_resample.py --shapes $SH --iters 10 > $O/rs_v$v.$RANDOM.jsonl 2>> $O/rs.err
done
cat $O/rs_v*.jsonl
FK_OP_V2=1 FK_OP_PRED_BACK=0 timeout 200 python tools/bench_resample.py --shapes $SH --iters 10 > $O/rs_v1_nopredict.jsonl 2>> $O/rs.err; cat $O/rs_v1_nopredict.jsonl
cd /tmp
for v in 0 1; do
  FK_OP_V2=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rs_stats_v$v -- python $R/tools/bench_resample.py --shapes $SH --iters 10 > $O/rs_under_stats_v$v.jsonl 2> $O/rs_stats_v$v.err
  python $R/tools/kernel_trace_summary.py $O/rs_stats_v$v > $O/kernel_durations_v$v.txt 2>&1; grep -i "onepass" $O/kernel_durations_v$v.txt | cut -c1-200
done
cd $R
for v in 0 1; do
  FK_OP_V2=$v timeout 300 python tools/op_phase.py --run --shapes 125x8000000,8x8000000,1x8000000 --iters 3 >> $O/op_phase.jsonl 2>> $O/op_phase.err
done
cat $O/op_phase.jsonl | cut -c1-1500
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*.db" -delete
