#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu.log
for lay in soa aos; do
  timeout 300 python bench.py --steps 10 --warmup 2 --layout $lay --no-cpu > gpurun_out/bench_${lay}.json 2> gpurun_out/bench_${lay}.err; echo "bench $lay rc=$?"
  python -c "import json;d=json.load(open('gpurun_out/bench_${lay}.json'));print('$lay', '%.3e'%d['value'], 'ms/step %.3f'%d['ms_per_step'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'frac %.3f'%d['roofline']['frac'])"
done
