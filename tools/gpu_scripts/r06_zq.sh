#!/bin/bash
# Round 6, lease zq: SQ / LDS counters of the several-lane fused UKF (ukf_mlg) and its smoother at dim_x 10..16
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06zq
mkdir -p $O
export TMPDIR=/tmp
cd $R
CMD="python tools/bench_configs.py --configs u --layouts soa"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set --output-format csv -d $O/pmc$i -- $CMD > /dev/null 2> $O/pmc$i.err; echo "pmc$i rc=$?"
done
python - <<PY > $O/pmc_rows.txt
import csv, glob, collections
for i in range(1, 4):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/pmc%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            if "ukf_mlg" in r["Kernel_Name"]:
                rows[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(rows.items()):
        print(i, k, {n: round(sum(v) / len(v)) for n, v in c.items()}, len(next(iter(c.values()))))
PY
cut -c1-420 $O/pmc_rows.txt
rm -rf $O/pmc1 $O/pmc2 $O/pmc3
