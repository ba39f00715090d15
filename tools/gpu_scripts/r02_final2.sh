#!/bin/bash
# Round 2, closing lease (second half of the round): every kernel of DESIGN section 5 once more, the chunking A/B in the
# same lease, then exactly what the driver does at round end (tools/gpu_scripts/round_check.sh).
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r02final2
R=$GRAFT_REPO_ROOT
mkdir -p $O
export TMPDIR=/tmp
rocm-smi --showclocks --showpower > $O/smi_before.txt 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg -- python $R/tools/bench_configs.py --configs 3456789ab --layouts soa,aos > $O/prof_cfg.log 2>&1; echo "cfg rc=$?"
cd $R
grep -E "^\{" $O/prof_cfg.log > $O/configs_all.jsonl; wc -l $O/configs_all.jsonl
find $O -name "*kernel_trace.csv" -size +1M -delete
for e in "1,1" "default" "1,1" "default"; do
  if [ "$e" = "default" ]; then unset FK_ML_CHUNKS; else export FK_ML_CHUNKS=$e; fi
  timeout 200 python tools/bench_configs.py --configs 3 2>/dev/null | sed "s/^{/{\"FK_ML_CHUNKS\": \"$e\", /" >> $O/c3_chunking.jsonl
done
unset FK_ML_CHUNKS
ML_WHAT=var timeout 200 python tools/exp_ml.py > $O/c3_variants.jsonl 2>/dev/null
bash tools/gpu_scripts/round_check.sh
