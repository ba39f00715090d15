#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
timeout 1200 python tools/bench_configs.py --configs ${CONFIGS:-4,5} --layouts ${LAYOUTS:-soa} > gpurun_out/configs.jsonl 2> gpurun_out/configs.err; echo "configs rc=$?"; tail -5 gpurun_out/configs.err
python - <<'PY'
import json
for l in open('gpurun_out/configs.jsonl'):
    d=json.loads(l); print("%-62s %9.3f ms  %.3e %s/s  %7.1f GB/s  %.3f  %s" % (d['kernel'], d['ms'], d['units_per_s'], d['unit'], d['achieved_GBs'], d['frac_of_8TBs'], {k:v for k,v in d.items() if k in ('parity_max_rel','bit_exact')}))
PY
