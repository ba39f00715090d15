#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
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
for i in 1 2 3; do for lay in soa aos; do
  timeout 300 python bench.py --steps 10 --warmup 2 --layout $lay --no-cpu > gpurun_out/bench_${lay}_$i.json 2> gpurun_out/bench_${lay}_$i.err
  python -c "import json;d=json.load(open('gpurun_out/bench_${lay}_$i.json'));print('C2 $lay run$i', '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'frac %.3f'%d['roofline']['frac'])"
done; done
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python $R/bench.py --steps 5 --warmup 1 --no-cpu > $R/gpurun_out/prof_stats.log 2>&1; echo "rocprof stats rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c5 -- python $R/tools/bench_configs.py --configs 5 > $R/gpurun_out/prof_c5.log 2>&1; echo "rocprof c5 rc=$?"
cd $R
for f in $(find gpurun_out/prof_stats -name "*kernel_stats.csv"); do cut -c1-140 $f | head -4; done
for f in $(find gpurun_out/prof_c5 -name "*kernel_stats.csv"); do cut -c1-200 $f | grep -E "fk::|Name" | head -12; done
