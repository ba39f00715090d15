#!/bin/bash
# hooks + kf_ml VAR family: parity, then timing
mkdir -p gpurun_out/r02v
timeout 900 python -m pytest tests/test_gpu_ukf_hooks.py tests/test_gpu_kf.py tests/test_gpu_ukf_device.py tests/test_gpu_ukf.py tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r02v/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02v/pytest.log
tail -15 gpurun_out/r02v/pytest.log
ML_WHAT=var timeout 300 python tools/exp_ml.py > gpurun_out/r02v/c3_variants.jsonl 2> gpurun_out/r02v/c3_variants.err
cat gpurun_out/r02v/c3_variants.jsonl
tail -3 gpurun_out/r02v/c3_variants.err
timeout 200 python tools/exp_ml.py > gpurun_out/r02v/c3_plain.jsonl 2>&1
head -2 gpurun_out/r02v/c3_plain.jsonl
