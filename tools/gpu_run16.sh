#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
timeout 900 python tools/bench_configs.py --configs 4 --layouts soa,aos > gpurun_out/configs4.jsonl 2> gpurun_out/configs4.err; tail -3 gpurun_out/configs4.err
python -c "
import json
for l in open('gpurun_out/configs4.jsonl'):
    d=json.loads(l); print('%-62s %9.3f ms  %.3e /s  %.3f %s' % (d['kernel'], d['ms'], d['units_per_s'], d['frac_of_8TBs'], d.get('parity_max_rel')))
"
