#!/bin/bash
# Round 6, lease zd: SQ counters of the one-lane-per-filter IMM kernel (who is busy: VALU, LDS, waiting)
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zd
mkdir -p $O
export TMPDIR=/tmp
cd $R
CMD="python tools/bench_configs.py --configs r --layouts soa"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_IFETCH SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/pmc$i -- $CMD > /dev/null 2> $O/pmc$i.err; echo "pmc$i rc=$?"
done
python tools/pmc_summary.py --kernel imm_lanes $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt
tail -2 $O/pmc1.err
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4
