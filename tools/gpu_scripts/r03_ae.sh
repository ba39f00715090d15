#!/bin/bash
# Round 3, lease ae: SQ counters of the fused UKF kernels of the closing build at C4 (two passes, counters only).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03ae
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python $R/tools/bench_ukf.py --dims 6x3 --layouts soa > /dev/null 2> $O/sq.err; echo "sq rc=$?"
timeout 200 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O/sq2 -- python $R/tools/bench_ukf.py --dims 6x3 --layouts soa > /dev/null 2> $O/sq2.err; echo "sq2 rc=$?"
cd $R
python tools/pmc_summary.py --all $O/sq $O/sq2 > $O/ukf_sq_counters.jsonl 2>&1; cut -c1-700 $O/ukf_sq_counters.jsonl
find $O -name "*counter_collection.csv" -delete
