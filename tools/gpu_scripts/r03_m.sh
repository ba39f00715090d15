#!/bin/bash
# lease: wave-cooperative NumPy-order sigma-point / unscented-transform kernels: parity, then A/B against round 2's per-lane pairs
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03m
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ukf.py tests/test_gpu_ukf_dims.py tests/test_gpu_tails.py tests/test_gpu_ukf_device.py tests/test_gpu_ukf_hooks.py tests/test_gpu_baseline_configs.py tests/test_gpu_api.py -m gpu -q -x -p no:cacheprovider > $O/pytest_ukf.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_ukf.log
for c in 1 0; do FK_UT_COOP=$c timeout 300 python tools/bench_configs.py --configs 4 --layouts aos 2>/dev/null | grep -E "sigma|unscented" | sed "s/^{/{\"FK_UT_COOP\": $c, /" >> $O/ut_coop.jsonl; done
cut -c1-260 $O/ut_coop.jsonl
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_default.json 2>/dev/null; python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r03m", "bench_default.json")))
print(d["value"], d["roofline"]["kernel_ms"], round(d["roofline"]["frac"], 4), d["hbm_probes"], d["gpu_clocks"].get("asic_serial"), d["gpu_clocks"].get("oam_id"), d["under_load"])
PY
