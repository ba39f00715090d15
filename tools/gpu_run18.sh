#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/bench_c5.py > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err; echo "c5 rc=$?"; cat gpurun_out/bench_c5.json; tail -2 gpurun_out/bench_c5.err
timeout 300 python tools/bench_c5.py --filters 125 --steps 10 > gpurun_out/bench_c5_125.json 2> gpurun_out/bench_c5_125.err; echo "c5-125 rc=$?"; cat gpurun_out/bench_c5_125.json
timeout 600 python tools/bench_c5.py --filters 125 --particles 8000000 --steps 3 --warmup 1 > gpurun_out/bench_c5_big.json 2> gpurun_out/bench_c5_big.err; echo "c5big rc=$?"; cat gpurun_out/bench_c5_big.json; tail -2 gpurun_out/bench_c5_big.err
