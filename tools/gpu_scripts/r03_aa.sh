#!/bin/bash
# Round 3, lease aa: tail filling of the fused UKF (FK_UKF_CHUNKS="G,H") with the round-3 step, A/B in one lease; SQ counters of
# the forward kernel at C4.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03aa
mkdir -p $O
export TMPDIR=/tmp
cd $R
B="timeout 200 python tools/bench_ukf.py --dims 6x3"
$B > $O/chunks_none.jsonl 2>/dev/null
for c in 2,2 2,4 3,4 2,8 4,4; do FK_UKF_CHUNKS=$c $B > $O/chunks_$c.jsonl 2>/dev/null; done
$B > $O/chunks_none_again.jsonl 2>/dev/null
cat $O/chunks_*.jsonl | grep "UKF (" | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print('%-44s %7.3f ms  frac %.3f  %s' % (r['kernel'], r['ms'], r['frac_of_8TBs'], r['switches']))
"
cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python $R/tools/bench_ukf.py --dims 6x3 --layouts soa > /dev/null 2> $O/sq.err; echo "sq rc=$?"
timeout 300 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA --output-format csv -d $O/sq2 -- python $R/tools/bench_ukf.py --dims 6x3 --layouts soa > /dev/null 2> $O/sq2.err; echo "sq2 rc=$?"
cd $R
python tools/pmc_summary.py --all $O/sq $O/sq2 > $O/ukf_sq_counters.txt 2>&1; grep -A3 "ukf_linear" $O/ukf_sq_counters.txt | cut -c1-600 | head -30
find $O -name "*counter_collection.csv" -size +1M -delete
