#!/bin/bash
# Round 6, lease zo: SQ / LDS counters of imm_quad_kernel<16,8,2> (outputs off and on)
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zo
mkdir -p $O
export TMPDIR=/tmp
cd $R
CMD="python tools/bench_imm_outputs.py --dims 16x8x2 --layout soa --iters 2"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL" "SQ_WAVE_CYCLES SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_DEP_WAIT SQ_EXP_REQ_FIFO_FULL"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/pmc$i -- $CMD > /dev/null 2> $O/pmc$i.err; echo "pmc$i rc=$?"
done
python - <<PY > $O/pmc_rows.txt
import csv, glob, collections
for i in range(1, 6):
    rows = collections.defaultdict(list)
    for f in glob.glob("$O/pmc%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            if "imm_quad" in r["Kernel_Name"]:
                rows[(r["Dispatch_Id"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    disp = sorted({k[0] for k in rows}, key=int)
    for d in disp:
        print(i, d, {k[1]: sum(v) for k, v in rows.items() if k[0] == d})
PY
cat $O/pmc_rows.txt | cut -c1-400
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5
