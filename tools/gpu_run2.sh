#!/bin/bash
# GPU visit 2: parity tests, bench A/B of the fast-kernel variants, rocprofv3 kernel stats + HBM PMC.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
for lay in soa aos; do for v in 0 1 2; do
  FK_FAST_VARIANT=$v timeout 300 python bench.py --steps 10 --warmup 2 --layout $lay --no-cpu > gpurun_out/bench_${lay}_v$v.json 2> gpurun_out/bench_${lay}_v$v.err; echo "bench $lay v$v rc=$?"
  python -c "import json;d=json.load(open('gpurun_out/bench_${lay}_v$v.json'));print('$lay v$v', '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'GB/s %.0f'%d['roofline']['achieved'], 'frac %.3f'%d['roofline']['frac'])"
done; done
timeout 300 python bench.py --steps 10 --warmup 2 --cpu-seconds 10 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cat gpurun_out/bench_default.json
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python $R/bench.py --steps 5 --warmup 1 --no-cpu > $R/gpurun_out/prof_stats.log 2>&1; echo "rocprof stats rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/prof_fetch.log 2>&1; echo "rocprof fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/prof_write.log 2>&1; echo "rocprof write rc=$?"
cd $R
find gpurun_out -name "*.csv" | head -20
for f in $(find gpurun_out/prof_stats -name "*kernel_stats.csv"); do head -6 $f; done
python tools/pmc_summary.py gpurun_out/prof_fetch gpurun_out/prof_write 2>&1 | tail -20
