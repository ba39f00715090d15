#!/bin/bash
# Round 5: the look-back window of resample_onepass_kernel (FK_OP_LOOKBACK: lanes of wave 0 that load a predecessor's hand-off
# words per look-back step; 64 shipped) at 32 and 16 -- the "7 % look-back over-fetch" lever: parity of the one-pass routes on each
# build, then kernel durations, shipped / 32 / 16 / shipped in one lease.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05m
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
RS="python $R/tools/bench_resample.py --shapes 125x8000000,8x8000000,1x8000000,1000x100000 --iters 10"
run() {
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rs_$1 -- $RS > $O/resample_$1.jsonl 2> $O/rs_$1.err
    python $R/tools/kernel_trace_summary.py $O/rs_$1 | grep onepass | sed "s/^/$1 /" | cut -c1-180 | tee -a $O/kernel_durations.txt
}
cp $R/filterpy_amd/libfilterhip.so /tmp/libfilterhip_shipped.so
run lookback64_a
for lb in 32 16; do
    cp $R/filterpy_amd/csrc/exp_build/libfilterhip_lb$lb.so $R/filterpy_amd/libfilterhip.so
    (cd $R && timeout 600 python -m pytest tests/test_gpu_resample.py -m gpu -q -p no:cacheprovider -k "onepass" 2>&1 | tail -1 | cut -c1-120)
    run lookback$lb
done
cp /tmp/libfilterhip_shipped.so $R/filterpy_amd/libfilterhip.so
run lookback64_b
find $O -name "*kernel_trace.csv" -size +1M -delete
