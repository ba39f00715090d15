#!/bin/bash
# Round 3, first lease: box state, the whole -m gpu suite (with the new UKF dims 7..16 cases), smoke, the default bench
# line (CPU sweep + streaming probes), and the RCCL branch executed once on a 1-rank nccl group (VERDICT r2 missing 2).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_scripts/r03_a.sh'
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03a
mkdir -p $O
export TMPDIR=/tmp
cd $R
bash tools/gpu_scripts/box_state.sh > $O/box_state.txt 2>&1
FK_PARITY_LOG=$O/parity_errors.jsonl timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-600 $O/bench_default.json
timeout 300 python bench.py --steps 10 --warmup 3 --force-dist --no-cpu > $O/bench_force_dist.json 2> $O/bench_force_dist.err; echo "force-dist rc=$?"; tail -3 $O/bench_force_dist.err; cut -c1-300 $O/bench_force_dist.json
timeout 300 python tools/bench_c5.py --filters 125 --particles 8000 --force-dist > $O/bench_c5_force_dist.json 2> $O/bench_c5_force_dist.err; echo "c5 force-dist rc=$?"; tail -3 $O/bench_c5_force_dist.err; cat $O/bench_c5_force_dist.json
# the driver's own launch line with ONE rank: torchrun sets WORLD_SIZE=1 -> bench must run (no group needed)
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; echo "torchrun1 rc=$?"; cut -c1-200 $O/bench_torchrun1.json
