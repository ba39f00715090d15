#!/bin/bash
# Round 4, lease y: the measurement DMA of the four-lane forward kernel (dims 10..16): one named test with its full trace
# first (the box before it failed that test in 2 s, a test whose kernels the change does not touch), then the whole GPU
# suite and the dims 10..16 forward rows.
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04y
mkdir -p $O
cd $R
rocm-smi --showuse --showmemuse 2>&1 | grep -E "GPU\[|use" | head -6 > $O/box.txt
if ! timeout 120 python -m pytest tests/test_gpu_api.py -m gpu -q -x -k config1 -p no:cacheprovider > $O/first.log 2>&1; then
    tail -40 $O/first.log | cut -c1-220
    HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 timeout 120 python -m pytest tests/test_gpu_api.py -m gpu -q -x -k config1 -p no:cacheprovider 2>&1 | grep -E "Error|error|\.py:[0-9]+" | head -30 | cut -c1-220
    dmesg 2>/dev/null | tail -5
    exit 1
fi
tail -1 $O/first.log
timeout 300 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1
tail -3 $O/pytest_gpu_full.log | cut -c1-200
cd /tmp
timeout 100 python $R/tools/bench_configs.py --configs b --layouts soa,aos 2>$O/bench_b.err | grep batch_filter | tee $O/bench_b.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['kernel'][:44], 'ms=%.3f'%d['ms'], 'frac=%.3f'%d['frac_of_8TBs'], d.get('parity_max_rel'))
"
tail -3 $O/bench_b.err | cut -c1-200
