#!/bin/bash
# Round 3, lease ac: the quad broadcast of the three-lane dim-9 kernels through ds_swizzle_b32 (LDS crossbar) instead of v_mov DPP
# (build-time -DFK_QUAD_SWIZZLE, variant library csrc/exp_build/libfilterhip_swz.so) against the shipped build, interleaved.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03ac
mkdir -p $O
cd $R
for i in 1 2; do
  timeout 200 python tools/bench_configs.py --configs 3 --layouts soa,aos > $O/c3_dpp_$i.jsonl 2>/dev/null
  FK_BENCH_LIB=filterpy_amd/csrc/exp_build/libfilterhip_swz.so timeout 200 python tools/bench_configs.py --configs 3 --layouts soa,aos > $O/c3_swz_$i.jsonl 2>/dev/null
done
python - <<'PY'
import json, os, glob
for f in sorted(glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03ac/c3_*.jsonl"))):
    for l in open(f):
        r = json.loads(l)
        print("%-14s %-44s %7.3f ms  frac %.3f  par %s" % (os.path.basename(f)[:-6], r["kernel"], r["ms"], r["frac_of_8TBs"], r.get("parity_max_rel")))
PY
