#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
timeout 900 python tools/bench_configs.py --configs 7 --layouts soa,aos > gpurun_out/configs7.jsonl 2> gpurun_out/configs7.err; tail -3 gpurun_out/configs7.err
python -c "
import json
for l in open('gpurun_out/configs7.jsonl'):
    d=json.loads(l); print('%-72s %9.3f ms  %.3e /s  %.3f' % (d['kernel'], d['ms'], d['units_per_s'], d['frac_of_8TBs']))
"
