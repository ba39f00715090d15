#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c5 -- python $R/tools/bench_configs.py --configs 5 > $R/gpurun_out/prof_c5.log 2>&1; echo "rocprof c5 rc=$?"
cd $R
for f in $(find gpurun_out/prof_c5 -name "*kernel_stats.csv"); do cut -c1-200 $f | grep -E "fk::|Name" | head -12; done
grep -E "^\{" gpurun_out/prof_c5.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('%-62s %9.3f ms  %.3e /s  %.3f %s' % (d['kernel'], d['ms'], d['units_per_s'], d['frac_of_8TBs'], d.get('bit_exact')))"
