#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
for rep in 1 2; do for lay in aos soa; do for v in 0 1 2 3 4; do
  FK_FAST_VARIANT=$v timeout 300 python bench.py --steps 10 --warmup 2 --layout $lay --no-cpu > gpurun_out/bench_${lay}_v$v.json 2> gpurun_out/bench_${lay}_v$v.err
  python -c "import json;d=json.load(open('gpurun_out/bench_${lay}_v$v.json'));print('C2 rep$rep $lay v$v', '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'frac %.3f'%d['roofline']['frac'], 'parity %.1e'%d['parity_max_rel_vs_oracle'])"
done; done; done
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -4
