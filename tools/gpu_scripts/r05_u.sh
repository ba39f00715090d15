#!/bin/bash
# Round 5: the four-lane smoother at dim_x 8 in NumPy order (the default there) carries 204 B of scratch under its three-waves launch
# bound; exp_build/libfilterhip_F.so has it bounded for two (234 VGPRs, none).  The smoother tests on F, then A/B/A/B.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05u
mkdir -p $O
cd $R
cp filterpy_amd/libfilterhip.so /tmp/libfilterhip_shipped.so
cp exp_build/libfilterhip_F.so filterpy_amd/libfilterhip.so
timeout 100 python -m pytest tests/test_gpu_kf.py -m gpu -q -p no:cacheprovider -k "rts or smoother or tail" 2>&1 | tail -2 | cut -c1-200 | tee $O/pytest_on_F.txt
cp /tmp/libfilterhip_shipped.so filterpy_amd/libfilterhip.so
cd /tmp
for lib in shipped F shipped F; do
    L=""; [ $lib != shipped ] && L=$R/exp_build/libfilterhip_$lib.so
    FK_LIB=$L KF_DIMS=8x4 timeout 60 python $R/tools/bench_configs.py --configs a --layouts aos 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); d['lib']='$lib'; print(json.dumps(d))
" | tee -a $O/rmlg8_waves_ab.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$lib', d['kernel'][:60], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d['frac_of_8TBs'], d.get('parity_max_rel'))
"
done
