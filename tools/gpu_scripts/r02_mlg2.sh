#!/bin/bash
mkdir -p gpurun_out/r02m
timeout 600 python -m pytest tests/test_gpu_kf.py -x -q -m gpu -k "smoother" > gpurun_out/r02m/pytest2.log 2>&1
tail -5 gpurun_out/r02m/pytest2.log
for L in 4 8; do echo "FK_RTS_LANES=$L"; FK_RTS_LANES=$L RTS_DIMS=13,14,15,16 timeout 200 python tools/exp_rts_mlg.py 2>/dev/null; done
