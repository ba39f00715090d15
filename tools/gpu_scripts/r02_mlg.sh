#!/bin/bash
mkdir -p gpurun_out/r02m
timeout 900 python -m pytest tests/test_gpu_kf.py tests/test_gpu_edges.py -x -q -m gpu -k "four_lane or padded or dims or edge or rts" > gpurun_out/r02m/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02m/pytest.log
tail -12 gpurun_out/r02m/pytest.log
timeout 300 python tools/bench_configs.py --configs b > gpurun_out/r02m/dims_10_16_b.jsonl 2> gpurun_out/r02m/dims.err
cut -c1-330 gpurun_out/r02m/dims_10_16_b.jsonl
tail -3 gpurun_out/r02m/dims.err
