#!/bin/bash
# Round 3, lease 3: the whole-vector resampling kernel after the per-thread fast path (bit-exactness, then KERNEL durations
# from rocprofv3 --kernel-trace --stats: a Python launch loop is host-bound below ~15 us per call), and the one-pass kernel
# with static chunk assignment (FK_OP_STATIC=1) against tickets.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c
mkdir -p $O
export TMPDIR=/tmp
cd $R
bash tools/gpu_scripts/box_state.sh > $O/box_state.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_resample.py -m gpu -q -x -p no:cacheprovider > $O/pytest_resample.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_resample.log
FK_OP_STATIC=1 timeout 600 python -m pytest tests/test_gpu_resample.py -m gpu -q -x -p no:cacheprovider -k "onepass or huge or c5_multi or chunk_parallel or goldens" > $O/pytest_resample_static.log 2>&1; echo "pytest static rc=$?"; tail -3 $O/pytest_resample_static.log
SH="--shapes 125x8000,1000x8000,125x4000,1000x2000,4000x8000,250x8000,500x8000 --iters 20"
cd /tmp
for v in eu4 eu8 local; do
  case $v in eu4) export FK_WHOLE_EU=4; unset FK_RESAMPLE_PATH;; eu8) export FK_WHOLE_EU=8; unset FK_RESAMPLE_PATH;; local) unset FK_WHOLE_EU; export FK_RESAMPLE_PATH=local;; esac
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python $R/tools/bench_resample.py $SH > $O/resample_$v.jsonl 2> $O/prof_$v.err; echo "$v rc=$?"
done
unset FK_WHOLE_EU FK_RESAMPLE_PATH
cd $R
python - <<'PY'
import csv, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r03c")
for v in ("eu4", "eu8", "local"):
    for f in glob.glob(os.path.join(O, "prof_" + v, "**", "*kernel_trace.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        # group consecutive launches of the resampling kernels by grid size (one group per shape)
        groups = {}
        for r in rows:
            name = r["Kernel_Name"]
            if "resample" not in name:
                continue
            key = (name[:60], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
            groups.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for key, d in groups.items():
            d = sorted(d)
            print(v, key, "n=%d median=%.2f us min=%.2f" % (len(d), d[len(d) // 2], d[0]))
PY
for st in 0 1; do FK_OP_STATIC=$st timeout 300 python tools/bench_resample.py --shapes 125x8000000,8x8000000,1x8000000,1000x100000 --iters 10 > $O/resample_long_static$st.jsonl 2>&1; echo "static=$st"; cat $O/resample_long_static$st.jsonl; done
find $O -name "*kernel_trace.csv" -size +3M -delete
