#!/bin/bash
# Round 3, lease 2: the new whole-vector resampling kernel (bit-exactness on every route, then timing: both register
# budgets, against round 2's local kernel in the same lease), the UKF dims 7..16 tests incl. fk_ukf_rts_correct_f64 at
# dim_x > 9, and the bench line on this box.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03b
mkdir -p $O
export TMPDIR=/tmp
cd $R
bash tools/gpu_scripts/box_state.sh > $O/box_state.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_resample.py tests/test_gpu_ukf_dims.py tests/test_gpu_ukf.py tests/test_gpu_ukf_device.py tests/test_gpu_ukf_hooks.py tests/test_gpu_tails.py -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
SH="--shapes 125x8000,1000x8000,125x4000,1000x2000,4000x8000 --iters 20"
for eu in 4 8; do FK_WHOLE_EU=$eu timeout 200 python tools/bench_resample.py $SH > $O/resample_whole_eu$eu.jsonl 2>&1; echo "eu=$eu"; cat $O/resample_whole_eu$eu.jsonl; done
timeout 200 python tools/bench_resample.py $SH > $O/resample_whole_default.jsonl 2>&1; cat $O/resample_whole_default.jsonl
FK_RESAMPLE_PATH=local timeout 200 python tools/bench_resample.py $SH > $O/resample_local.jsonl 2>&1; echo "local"; cat $O/resample_local.jsonl
timeout 200 python tools/bench_resample.py --shapes 125x8000000,8x8000000 --iters 10 > $O/resample_long.jsonl 2>&1; cat $O/resample_long.jsonl
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03b/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["hbm_probes"], d["gpu_clocks"].get("asic_serial"), d["gpu_clocks"].get("oam_id"))
print(d["cpu_baseline"]["sample"][-400:])
PY
