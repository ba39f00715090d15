#!/bin/bash
# Round 5: single-binade speculation in resample_onepass_kernel (where the guess interval lies inside one binade the second
# candidate's increments, tie test and wave scan are not formed) -- every route bit-exact, then A/B/A against the build that
# always speculates on two binades (csrc/exp_build/libfilterhip_two_binades.so swapped in place), kernel durations under rocprofv3.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05l
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_resample.py -m gpu -q -p no:cacheprovider -k "onepass or c5 or huge or chunk_parallel or bank_of_filters or python_api" > $O/tests_1.log 2>&1
tail -3 $O/tests_1.log | cut -c1-200
cd /tmp
export TMPDIR=/tmp
RS="python $R/tools/bench_resample.py --shapes 125x8000000,8x8000000,1x8000000,1000x100000,125x1000000 --iters 10"
run() {
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rs_$1 -- $RS > $O/resample_$1.jsonl 2> $O/rs_$1.err
    python $R/tools/kernel_trace_summary.py $O/rs_$1 | grep onepass | cut -c1-170 | tee $O/kernel_durations_$1.txt
    python -c "
import json
for l in open('$O/resample_$1.jsonl'):
    if l.startswith('{'):
        d=json.loads(l); print('$1', d['filters'], d['particles'], 'ms', d['ms'], 'frac', round(d['frac_hbm'],3))
"
}
run single_a
cp $R/filterpy_amd/libfilterhip.so /tmp/libfilterhip_shipped.so
cp $R/filterpy_amd/csrc/exp_build/libfilterhip_two_binades.so $R/filterpy_amd/libfilterhip.so
run two_binades
cp /tmp/libfilterhip_shipped.so $R/filterpy_amd/libfilterhip.so
run single_b
find $O -name "*kernel_trace.csv" -size +1M -delete
