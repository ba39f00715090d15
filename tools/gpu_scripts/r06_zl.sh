#!/bin/bash
# Round 6, lease zl: SQ counters of the four-lane kernels at dim_x 16 (kf_mlg, rts_mlg, imm_quad): who is busy, who waits
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zl
mkdir -p $O
export TMPDIR=/tmp
cd $R
CMD="python tools/bench_configs.py --configs br --layouts soa"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_IFETCH SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set --output-format csv -d $O/pmc$i -- $CMD > /dev/null 2> $O/pmc$i.err; echo "pmc$i rc=$?"
done
for k in kf_mlg_kernel rts_mlg_kernel rts_mlx_kernel imm_quad; do
  echo "=== $k" >> $O/pmc_summary.txt
  python tools/pmc_summary.py --kernel $k $O/pmc1 $O/pmc2 $O/pmc3 >> $O/pmc_summary.txt 2>&1
done
cat $O/pmc_summary.txt | cut -c1-250
rm -rf $O/pmc1 $O/pmc2 $O/pmc3
