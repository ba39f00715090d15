#!/bin/bash
# lease: one-pass kernel with speculation: exactness in all three protocols, then timing A/B + phase clocks
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03i}
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_resample.py -m gpu -q -x -p no:cacheprovider > $O/pytest_resample.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_resample.log
SH="--shapes 125x8000000,8x8000000,1x8000000,1000x100000,125x1000000 --iters 10"
for m in spec twostage tickets; do
  case $m in spec) unset FK_OP_SPEC FK_OP_STATIC;; twostage) export FK_OP_SPEC=0; unset FK_OP_STATIC;; tickets) export FK_OP_SPEC=0 FK_OP_STATIC=0;; esac
  timeout 300 python tools/bench_resample.py $SH > $O/resample_long_$m.jsonl 2>&1; echo "== $m"; grep -v amdgpu $O/resample_long_$m.jsonl
done
unset FK_OP_SPEC FK_OP_STATIC
python tools/op_phase.py --run --shapes 125x8000000 --iters 3 > $O/onepass_phase_clocks.jsonl 2>&1; cat $O/onepass_phase_clocks.jsonl | cut -c1-900
FK_OP_SPEC=0 python tools/op_phase.py --run --shapes 125x8000000 --iters 3 >> $O/onepass_phase_clocks.jsonl 2>&1; tail -1 $O/onepass_phase_clocks.jsonl | cut -c1-900
