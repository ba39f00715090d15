#!/bin/bash
# Round 5, second lease: (1) parity of what changed since r05_a -- the persistent grid of resample_whole_kernel, the explicit
# vmcnt drain in kf_ml's persistent hand-over (status = NULL through the ABI), the UKF routing (several-lane kernels on by default,
# smoother from dim_x 7), the last-epoch replay; (2) C5 timings (kernel durations under rocprofv3 + per-call from Python);
# (3) fused vs split UKF at dim_x >= 10 through the Python class (FK_UKF_MLG=0 = the building blocks); (4) the smoother at 7..9
# one lane vs several (FK_UKF_MLG_RTS_MIN_NX).
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05c
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_resample.py tests/test_gpu_ukf_mlg.py tests/test_gpu_api.py -m gpu -q -x -p no:cacheprovider > $O/tests_1.log 2>&1
tail -5 $O/tests_1.log | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_kf.py -m gpu -q -x -p no:cacheprovider -k "persistent or placement or interleav" > $O/tests_2.log 2>&1
tail -3 $O/tests_2.log | cut -c1-200
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "ukf or UKF" > $O/tests_3.log 2>&1
tail -3 $O/tests_3.log | cut -c1-200
cd /tmp
export TMPDIR=/tmp
# C5: kernel durations (rocprofv3) of the resampling shapes, grid forced to one workgroup per filter next to the default
for g in default 0; do
    if [ $g = 0 ]; then export FK_WHOLE_GRID=0; else unset FK_WHOLE_GRID; fi
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5_$g -- python $R/tools/bench_configs.py --configs 5 > $O/c5_$g.jsonl 2>$O/c5_$g.err
    python $R/tools/kernel_trace_summary.py $O/prof_c5_$g 2>/dev/null | grep -i "resample" | head -12 > $O/c5_kernels_$g.txt
    echo "== grid $g"; cat $O/c5_kernels_$g.txt | cut -c1-200
    grep -h "systematic" $O/c5_$g.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['kernel'][:60], 'ms=%.4f'%d['ms'], 'frac=%.3f'%d['frac_of_8TBs'], d.get('bit_exact'))
"
done
unset FK_WHOLE_GRID
find $O -name "*.db" -delete 2>/dev/null
# fused vs split through the class
timeout 600 python $R/tools/bench_ukf_class.py > $O/ukf_class.jsonl 2>$O/ukf_class.err
cat $O/ukf_class.jsonl | cut -c1-250
# smoother at dim_x 7..9: one lane (RTS_MIN_NX=10) vs several lanes (default 7)
for mn in 10 7; do
    FK_UKF_MLG_RTS_MIN_NX=$mn timeout 200 python $R/tools/bench_ukf.py --dims 7x3,8x4,9x3 --N 100000 --T 100 2>>$O/bench_small.err | grep smoother | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); d['rts_min_nx']=$mn; print(json.dumps(d))
" | tee -a $O/ukf_rts_small_ab.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('rts_min_nx', d['rts_min_nx'], d['kernel'][:48], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d.get('frac_of_8TBs',0), d.get('parity_max_rel'))
"
done
