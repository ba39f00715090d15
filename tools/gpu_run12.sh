#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do for lay in aos soa; do for x in 0 1; do
  FK_FAST_XCD=$x timeout 300 python bench.py --steps 10 --warmup 2 --layout $lay --no-cpu > gpurun_out/bench_${lay}_x$x.json 2> gpurun_out/bench_${lay}_x$x.err
  python -c "import json;d=json.load(open('gpurun_out/bench_${lay}_x$x.json'));print('C2 rep$rep $lay xcd=$x', '%.3e'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'frac %.3f'%d['roofline']['frac'], 'parity %.1e'%d['parity_max_rel_vs_oracle'])"
done; done; done
python - <<'PY'
import torch, time
d=torch.device('cuda')
n=2*1024**3//8
a=torch.empty(n,dtype=torch.float64,device=d)
for _ in range(3): a.fill_(1.0)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): a.fill_(1.0)
torch.cuda.synchronize(); print("fill GB/s", n*8/((time.perf_counter()-t)/10)/1e9)
PY
timeout 600 python -m pytest tests/test_gpu_kf.py -m gpu -q -x 2>&1 | tail -2
FK_FAST_XCD=1 timeout 600 python -m pytest tests/test_gpu_kf.py -m gpu -q -x 2>&1 | tail -2
