#!/bin/bash
# Round 3, lease ad: tail filling of the fused UKF smoother (FK_UKF_RTS_CHUNKS): bit-identity + golden tests, then one launch against
# the default policy and other decompositions at BASELINE configs[3] (1e5 tracks x 100 steps), interleaved.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03ad
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ukf_dims.py tests/test_gpu_ukf.py tests/test_gpu_variants.py -m gpu -q -x -p no:cacheprovider -k "smoother or rts" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
B="timeout 200 python tools/bench_ukf.py --dims 6x3,5x2"
for c in 1,1 default 3,4 2,4 3,8 4,4 1,1 default; do
  if [ $c = default ]; then $B > $O/tmp.jsonl 2>/dev/null; else FK_UKF_RTS_CHUNKS=$c $B > $O/tmp.jsonl 2>/dev/null; fi
  grep smoother $O/tmp.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$c', '%-40s %7.3f ms frac %.3f par %.1e' % (r['kernel'], r['ms'], r['frac_of_8TBs'], r['parity_max_rel']))
" | tee -a $O/chunks.txt
done
