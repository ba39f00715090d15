#!/bin/bash
# rocprofv3 evidence for profiles/: kernel stats of the default bench, HBM PMC counters in separate
# passes (FETCH_SIZE, WRITE_SIZE: never combined with trace domains), kernel stats of the other configs.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_scripts/profile.sh'
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python $R/bench.py --steps 5 --warmup 1 --no-cpu > $R/gpurun_out/prof_stats.log 2>&1; echo "stats rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/prof_fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/prof_write.log 2>&1; echo "write rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg -- python $R/tools/bench_configs.py --configs 3456789a --layouts soa > $R/gpurun_out/prof_cfg.log 2>&1; echo "cfg rc=$?"
cd $R
for f in $(find gpurun_out/prof_stats -name "*kernel_stats.csv"); do cut -c1-160 $f | head -4; done
python tools/pmc_summary.py gpurun_out/prof_fetch gpurun_out/prof_write 2>&1 | tail -6
for f in $(find gpurun_out/prof_cfg -name "*kernel_stats.csv"); do cut -c1-200 $f | grep -E "fk::|Name" | head -30; done
grep -E "^\{" gpurun_out/prof_cfg.log > gpurun_out/configs_all.jsonl
