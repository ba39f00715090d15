#!/bin/bash
# What the driver does at round end, in one gpurun call:  pytest -m gpu, smoke(), default bench.
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/gpu_scripts/round_check.sh'
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
FK_PARITY_LOG=gpurun_out/parity_errors.jsonl timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; cat gpurun_out/bench_default.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu > gpurun_out/bench_torchrun.json 2> gpurun_out/bench_torchrun.err; echo "torchrun rc=$?"
