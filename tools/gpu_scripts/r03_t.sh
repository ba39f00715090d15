#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03t
mkdir -p $O
cd $R
WITH_MASK=1 timeout 300 python tools/debug/ex_aos_dims.py > $O/ex_aos_dims_mask.log 2>&1; cut -c1-330 $O/ex_aos_dims_mask.log | tail -30
timeout 600 python -m pytest tests/test_gpu_kf.py -m gpu -q -p no:cacheprovider -k "saver_histories" > $O/pytest_saver.log 2>&1; grep -E "^FAILED|passed|failed" $O/pytest_saver.log | cut -c1-200 | tail -40
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu > $O/bench_placed.json 2> $O/bench_placed.err; tail -2 $O/bench_placed.err; python -c "
import json; d = json.load(open('$O/bench_placed.json')); p = d['placement']; print(d['roofline']['kernel_ms'], d['roofline']['frac'], {k: v for k, v in p.items() if k != 'grid_ms'}); print(p.get('grid_ms'))"
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu --placement none > $O/bench_unplaced.json 2>/dev/null; python -c "
import json; d = json.load(open('$O/bench_unplaced.json')); print('unplaced', d['roofline']['kernel_ms'], d['roofline']['frac'])"
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu --layout soa > $O/bench_placed_soa.json 2>/dev/null; python -c "
import json; d = json.load(open('$O/bench_placed_soa.json')); p = d['placement']; print('soa', d['roofline']['kernel_ms'], d['roofline']['frac'], {k: v for k, v in p.items() if k != 'grid_ms'})"
