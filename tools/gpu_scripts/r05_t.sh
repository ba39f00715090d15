#!/bin/bash
# Round 5, last lease (5 GPU-minutes left): smoke() and bench.py's default run on the library as shipped after the per-shape work
# of r05_n .. r05_s (the closing evidence lease r05_final.sh ran before it; the BASELINE kernels' source has not changed since),
# and the Saver-history rows.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05t
mkdir -p $O
cd $R
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-200
timeout 170 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-600 $O/bench_default.json
cd /tmp
timeout 60 python $R/tools/bench_configs.py --configs e --layouts soa,aos 2>/dev/null | grep "^{" | grep -v generic > $O/extras_shipped.jsonl; python -c "
import sys,json
for l in open('$O/extras_shipped.jsonl'):
    d=json.loads(l); print(d['kernel'][:86], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d['frac_of_8TBs'])
"
