#!/bin/bash
# lease: Saver histories from kf_fast (parity + A/B against the generic kernel), wave-cooperative steady-state kernel,
# one-pass resampling with 12 / 16 weights per thread (A/B builds), SQ counters of the C3 / C4 kernels, the box probe
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03o
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kf.py tests/test_gpu_api.py tests/test_gpu_zz_saver.py tests/test_gpu_tails.py tests/test_gpu_variants.py tests/test_gpu_zz_module_steadystate.py -m gpu -q -x -p no:cacheprovider -k "saver or Saver or extras or steady or lean_fast or batch_filter_goldens or tuning" > $O/pytest_extras_steady.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_extras_steady.log
timeout 300 python tools/bench_configs.py --configs e9 2>/dev/null | grep "^{" > $O/extras_steady.jsonl; cut -c1-250 $O/extras_steady.jsonl
timeout 400 python tools/exp_rs_variants.py --run > $O/onepass_items.log 2>&1; tail -8 $O/onepass_items.log | cut -c1-600
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_default.json 2>/dev/null
bash tools/gpu_scripts/slow_box_probe.sh $O $O/bench_default.json 2>&1 | tail -60
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/c34_sq -- python $R/tools/bench_configs.py --configs 34 --layouts soa > /dev/null 2> $O/c34_sq.err
cd $R
python tools/pmc_summary.py $O/c34_sq --kernel kernel > $O/c34_sq_summary.txt 2>&1; cut -c1-700 $O/c34_sq_summary.txt | head -20; tail -2 $O/c34_sq.err
find $O -name "*counter_collection.csv" -size +1M -delete
