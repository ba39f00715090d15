#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02p
cd $R
RTS_DIMS=12,14,15,16 timeout 200 python tools/exp_rts_mlg.py 2>/dev/null
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_IFETCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | cut -d' ' -f1)
  RTS_DIMS=14,16 RTS_T=20 timeout 300 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/r02p/$tag -o pmc -- python tools/exp_rts_mlg.py > /dev/null 2>&1
done
python tools/pmc_summary.py --all gpurun_out/r02p | grep -i "rts_mlg" | cut -c40-200
