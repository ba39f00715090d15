#!/bin/bash
# Round 4, lease m: (1) wave_index(): the wave's slab descriptors scalar -> no waterfall loop around every slab store of the
# several-lanes-per-track kernels; (2) kf_ml XCH: quad exchanges through a wave-private LDS block instead of DPP moves.
# Full suite, C3 with FK_ML_XCH=1 / 0, then the C3 / dims 10..16 / extras rows under rocprofv3 stats.
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04m
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu_full.log | cut -c1-220
cd /tmp
for x in 1 0; do
  FK_ML_XCH=$x timeout 300 python $R/tools/bench_configs.py --configs 3 --layouts soa,aos > $O/c3_xch$x.jsonl 2> $O/c3_xch$x.err
  python - <<PY
import json
for l in open("$O/c3_xch$x.jsonl"):
    if l.startswith("{"):
        d=json.loads(l); print("XCH=$x", d["kernel"][:60], "ms=%.3f"%d["ms"], "frac=%.3f"%d["frac_of_8TBs"], d.get("parity_max_rel",""))
PY
done
FK_ML_CHUNKS=1,1 timeout 300 python $R/tools/bench_configs.py --configs 3 --layouts soa,aos 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('one launch', d['kernel'][:60], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d['frac_of_8TBs'])
"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg -- python $R/tools/bench_configs.py --configs 3be --layouts soa,aos > $O/prof_cfg.log 2>&1; echo "cfg rc=$?"
grep -E "^\{" $O/prof_cfg.log > $O/configs_3be.jsonl; python - <<PY
import json
for l in open("$O/configs_3be.jsonl"):
    d=json.loads(l); print(d["kernel"][:95], "ms=%.3f"%d["ms"], "frac=%.3f"%d["frac_of_8TBs"], d.get("parity_max_rel",""))
PY
grep -v "^{" $O/prof_cfg.log | grep -iE "error|assert|Traceback" | head
python $R/tools/kernel_trace_summary.py $O/prof_cfg > $O/configs_3be_kernel_durations.txt 2>&1
for f in $(find $O/prof_cfg -name "*kernel_stats.csv"); do cp $f $O/configs_3be_kernel_stats.csv; done
find $O -name "*kernel_trace.csv" -size +1M -delete
