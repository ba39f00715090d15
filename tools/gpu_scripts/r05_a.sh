#!/bin/bash
# Round 5, first lease: the several-lane UKF filter and smoother (csrc/ukf_mlg.hip, dims 10..16) have been checked on the host
# (tests/test_hostcheck_ukf_quad.py, tests/test_cpu_dryrun_ukf_mlg.py) and probed on a GPU for 3 seconds (profiles/r04/lease_q).  Their GPU parity tests, then -- if green -- the whole UKF suite with the kernels switched
# on, then their first timings next to the split path's.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
export FK_UKF_MLG=1
timeout 600 python -m pytest tests/test_gpu_ukf_mlg.py -m gpu -q -p no:cacheprovider > $O/ukf_mlg_tests.log 2>&1
tail -15 $O/ukf_mlg_tests.log | cut -c1-220
if grep -q " passed" $O/ukf_mlg_tests.log && ! grep -q "failed" $O/ukf_mlg_tests.log; then
    timeout 600 python -m pytest tests -m gpu -q -k "ukf or UKF" -p no:cacheprovider 2>&1 | tail -4 | cut -c1-220
    cd /tmp
    for d in 10x2 12x3 14x4 16x4; do
        timeout 200 python $R/tools/bench_ukf.py --dims $d --N 50000 --T 50 --dense 2>$O/bench_$d.err | tee -a $O/ukf_mlg_bench.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['kernel'][:48], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d.get('frac_of_8TBs',0), d.get('parity_max_rel'))
"
    done
    # the smoother at dim_x 13..16: eight lanes per track (the default there: 42-64 KB of code) against four (72-105 KB)
    for ln in 8 4; do
        FK_UKF_MLG_RTS_LANES=$ln timeout 200 python $R/tools/bench_ukf.py --dims 14x4,16x4 --N 50000 --T 50 --dense 2>>$O/bench_lanes.err | grep smoother | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); d['rts_lanes']=$ln; print(json.dumps(d))
" | tee -a $O/ukf_rts_lanes_ab.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('lanes', d['rts_lanes'], d['kernel'][:48], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d.get('frac_of_8TBs',0), d.get('parity_max_rel'))
"
    done
    # the one-lane classes at dim_x 8 / 9 (one wave per SIMD, scratch) against the four-lane kernels on the same calls
    FK_UKF_MLG_MIN_NX=7 timeout 300 python -m pytest $R/tests/test_gpu_ukf_mlg.py -m gpu -q -x -k "small_dims" -p no:cacheprovider 2>&1 | tail -2 | cut -c1-200
    for d in 8x4 9x3; do
        for mn in 10 7; do
            FK_UKF_MLG_MIN_NX=$mn timeout 200 python $R/tools/bench_ukf.py --dims $d --N 100000 --T 100 2>>$O/bench_small.err | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); d['min_nx']=$mn; print(json.dumps(d))
" | tee -a $O/ukf_small_ab.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('min_nx', d['min_nx'], d['kernel'][:48], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d.get('frac_of_8TBs',0), d.get('parity_max_rel'))
"
        done
    done
fi
