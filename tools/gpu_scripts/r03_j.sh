set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03j
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_resample.py -m gpu -q -x -p no:cacheprovider -k "onepass or huge or c5_multi or chunk_parallel or goldens or local_kernel" > $O/pytest_resample.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_resample.log
SH="--shapes 125x8000000,8x8000000,1000x100000,125x1000000 --iters 10"
for m in spec twostage; do
  case $m in spec) unset FK_OP_SPEC FK_OP_STATIC;; twostage) export FK_OP_SPEC=0; unset FK_OP_STATIC;; esac
  timeout 300 python tools/bench_resample.py $SH > $O/resample_long_$m.jsonl 2>&1; echo "== $m"; grep -v amdgpu $O/resample_long_$m.jsonl
done
