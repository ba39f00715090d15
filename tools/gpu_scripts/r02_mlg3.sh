#!/bin/bash
mkdir -p gpurun_out/r02m
timeout 900 python -m pytest tests/test_gpu_kf.py tests/test_gpu_edges.py -x -q -m gpu -k "four_lane or smoother or multilane or edge or padded" > gpurun_out/r02m/pytest3.log 2>&1
tail -6 gpurun_out/r02m/pytest3.log
