#!/bin/bash
# Round 6, sixth lease: what bounds the one-pass kernel?  (a) the slow path as a call (shipped) against inlined (exp_build variant),
# (b) time against the workgroups a CU holds (3 / 4 by LDS padding, 5 / 6 / 7 by instantiation), (c) SQ / TCC counters at 5 and 7.
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06f
mkdir -p $O
export TMPDIR=/tmp
cd $R
SH=125x8000000,1000x100000
INL=$R/filterpy_amd/csrc/exp_build/libfilterhip_inl.so
for env in "FK_OP_WAVES=5" "FK_LIB=$INL FK_OP_WAVES=5" "FK_LIB=$INL FK_OP_WAVES=6" "FK_LIB=$INL FK_OP_WAVES=7" "FK_OP_WAVES=7" "FK_OP_WAVES=5 FK_OP_LDS_PAD=17900" "FK_OP_WAVES=5 FK_OP_LDS_PAD=25000" "FK_OP_WAVES=5 FK_OP_LDS_PAD=40000" "FK_OP_V2=0" "FK_LIB=$INL FK_OP_WAVES=5" "FK_OP_WAVES=5"; do
  echo "== $env" >> $O/rs_ab.txt
  env $env timeout 200 python tools/bench_resample.py --shapes $SH --iters 10 >> $O/rs_ab.txt 2>> $O/rs.err
done
cat $O/rs_ab.txt | cut -c1-110
cd /tmp
for w in 5 7; do
 for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA" "TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ TCC_BUSY TCC_REQ" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"; do
  tag=w${w}_$(echo $set | cut -c1-14 | tr ' ' '_')
  FK_OP_WAVES=$w timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -- python $R/tools/bench_resample.py --shapes 125x8000000 --iters 4 > /dev/null 2> $O/pmc_$tag.err
  python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$O/pmc_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "onepass2" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("waves=$w", {k: round(sum(v)/len(v),1) for k,v in acc.items()}, "launches", max([len(v) for v in acc.values()] or [0]))
PY
 done
done
find $O -name "*counter_collection.csv" -size +1M -delete; find $O -name "*.db" -delete
