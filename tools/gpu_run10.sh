#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
for v in 0 1 2; do echo "== FK_FAST_VARIANT=$v"; FK_FAST_VARIANT=$v timeout 600 python tools/bench_configs.py --configs 36 --layouts soa,aos 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l.rstrip()); continue
    print('%-50s %9.3f ms  %.3e /s  %.3f  %s' % (d['kernel'], d['ms'], d['units_per_s'], d['frac_of_8TBs'], d.get('parity_max_rel')))
"; done
for lay in aos soa; do timeout 300 python bench.py --steps 10 --warmup 2 --layout $lay --no-cpu > gpurun_out/bench_${lay}.json 2> gpurun_out/bench_${lay}.err
  python -c "import json;d=json.load(open('gpurun_out/bench_${lay}.json'));print('C2 $lay', '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'frac %.3f'%d['roofline']['frac'])"; done
