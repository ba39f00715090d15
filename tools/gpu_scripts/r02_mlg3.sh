#!/bin/bash
mkdir -p gpurun_out/r02m
timeout 900 python -m pytest tests/test_gpu_kf.py tests/test_gpu_edges.py tests/test_gpu_baseline_configs.py -x -q -m gpu -k "four_lane or smoother or multilane or edge or padded or rts or c3" > gpurun_out/r02m/pytest3.log 2>&1
tail -6 gpurun_out/r02m/pytest3.log
timeout 300 python tools/bench_configs.py --configs b3 > gpurun_out/r02m/dims_10_16_c.jsonl 2> gpurun_out/r02m/dims.err
cut -c1-40,95-130,222-262 gpurun_out/r02m/dims_10_16_c.jsonl
