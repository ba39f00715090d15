#!/bin/bash
# First GPU visit: parity tests, smoke, short bench, HBM write/copy calibration, rocprofv3 kernel stats.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
ls /root/reference 2>&1 | head -2
rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -8
nproc
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
python - > gpurun_out/bw_calib.log 2>&1 <<'PY'
import torch, time
d=torch.device('cuda')
n=2*1024**3//8
a=torch.empty(n,dtype=torch.float64,device=d); b=torch.empty_like(a)
for name,fn,bytes_ in (("fill(write)",lambda: a.fill_(1.0), n*8),("copy(r+w)",lambda: b.copy_(a), 2*n*8)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
    print(name, bytes_/dt/1e9, "GB/s")
PY
cat gpurun_out/bw_calib.log
for lay in soa aos; do
  timeout 600 python bench.py --steps 10 --warmup 2 --layout $lay $( [ $lay = aos ] && echo --no-cpu ) > gpurun_out/bench_$lay.json 2> gpurun_out/bench_$lay.err; echo "bench $lay rc=$?"; cat gpurun_out/bench_$lay.json; tail -3 gpurun_out/bench_$lay.err
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_soa -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_soa.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; find gpurun_out/prof_soa -name "*stats*" | head; for f in $(find gpurun_out/prof_soa -name "*kernel_stats.csv"); do head -8 $f; done
