#!/bin/bash
# lease: whole-vector kernel after the merged boundary loop / trimmed chain: exactness, phase clocks, kernel durations
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03e}
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_resample.py -m gpu -q -x -p no:cacheprovider -k "whole or goldens or bank_of or garbage or local_kernel" > $O/pytest_resample.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_resample.log
python tools/op_phase.py --run --whole --shapes 125x8000,1000x8000,125x4000,1000x2000 --iters 5 > $O/whole_phase_clocks.jsonl 2>&1; cat $O/whole_phase_clocks.jsonl
python tools/op_phase.py --run --whole --stratified 1 --shapes 125x8000,1000x8000 --iters 5 >> $O/whole_phase_clocks.jsonl 2>&1; tail -2 $O/whole_phase_clocks.jsonl
SH="--shapes 125x8000,1000x8000,125x4000,1000x2000,4000x8000,250x8000,500x8000 --iters 20"
cd /tmp
for v in eu4 eu8; do
  case $v in eu4) export FK_WHOLE_EU=4;; eu8) export FK_WHOLE_EU=8;; esac
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python $R/tools/bench_resample.py $SH > $O/resample_$v.jsonl 2> $O/prof_$v.err; echo "$v rc=$?"
done
unset FK_WHOLE_EU
cd $R
python tools/kernel_trace_summary.py $O/prof_eu4 $O/prof_eu8 | tee $O/kernel_durations.txt
find $O -name "*kernel_trace.csv" -size +3M -delete
