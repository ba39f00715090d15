#!/bin/bash
# Round 4, first lease: the full GPU suite on the round's first batch of changes (pair-regrouped UKF sums, interleaved
# covariance histories, the one-pass resampler's repair pass, IMM chunking with a mask, the overlapped exchange), then
#   * fused UKF A/B: FK_UKF_PAIRED=0 (index-order sums) against the default (pair-regrouped), every class, both layouts
#   * bench.py: interleave (default) / none / probe, and the process-to-process spread of interleave against none
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_scripts/r04_a.sh'
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04a
mkdir -p $O
export TMPDIR=/tmp
cd $R
bash tools/gpu_scripts/box_state.sh > $O/box_state.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest_gpu_full.log
B="python $R/tools/bench_ukf.py --dims 6x3,4x2,2x2,8x4,9x3,9x4"
timeout 300 $B > $O/ukf_paired.jsonl 2> $O/ukf_paired.err; echo "paired rc=$?"
FK_UKF_PAIRED=0 timeout 300 $B > $O/ukf_index_order.jsonl 2> $O/ukf_index_order.err; echo "index rc=$?"
timeout 200 python tools/bench_ukf.py --dims 6x3 --N 1000000 --T 20 > $O/ukf_paired_1e6.jsonl 2>> $O/ukf_paired.err
cut -c1-230 $O/ukf_paired.jsonl; cut -c1-230 $O/ukf_index_order.jsonl; cut -c1-230 $O/ukf_paired_1e6.jsonl
BENCH="python $R/bench.py --steps 20 --warmup 5"
timeout 600 $BENCH > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-400 $O/bench_default.json
for i in 1 2 3; do
  timeout 300 $BENCH --no-cpu --placement none > $O/bench_none_$i.json 2>/dev/null
  timeout 300 $BENCH --no-cpu > $O/bench_interleave_$i.json 2>/dev/null
done
timeout 300 $BENCH --no-cpu --layout soa > $O/bench_interleave_soa.json 2>/dev/null
timeout 300 $BENCH --no-cpu --layout soa --placement none > $O/bench_none_soa.json 2>/dev/null
timeout 400 $BENCH --no-cpu --placement probe > $O/bench_probe.json 2>/dev/null
python - <<'PY'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04a"
for f in sorted(glob.glob(O + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "kernel_ms", round(d["roofline"]["kernel_ms"], 4), "ms_per_step", round(d["ms_per_step"], 4),
              "frac", round(d["roofline"]["frac"], 4), {k: v for k, v in d["placement"].items() if k != "grid_ms"})
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
