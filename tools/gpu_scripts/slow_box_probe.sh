#!/bin/bash
# Run at the end of a lease: if the default bench line of this box says the headline kernel took > 5.9 ms (the slow GPUs of
# the pool: 6.2-6.8 ms against 5.3-5.4), collect what could tell WHY -- the box state, the other layout, the other kernels.
#   bash tools/gpu_scripts/slow_box_probe.sh <out dir> <bench json of this lease>
O=$1; J=$2
R=$GRAFT_REPO_ROOT
cd $R
MS=$(python -c "import json,sys; print(json.load(open('$J'))['roofline']['kernel_ms'])" 2>/dev/null || echo 0)
echo "headline kernel on this box: $MS ms"
bash tools/gpu_scripts/box_state.sh > $O/box_state.txt 2>&1
# every box: the same counters and track counts, so that a slow and a fast box can be laid side by side
# is it the SHAPE (4,2) -- records of 32 / 128 bytes -- or the track count / the addresses?  other track counts, and (4,2) through bench_configs
for n in 999936 1048576 1003520 500000; do timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu --tracks $n 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'tracks': $n, 'kernel_ms': d['roofline']['kernel_ms'], 'frac': d['roofline']['frac']}))" >> $O/slow_tracks.jsonl; done
cat $O/slow_tracks.jsonl
# the same workload with the output arrays placed differently in HBM, with the 3-wave variant, with the XCD swizzle
for kv in FK_BENCH_PAD_MB=0 FK_BENCH_PAD_MB=37 FK_BENCH_PAD_MB=1001 FK_FAST_VARIANT=1 FK_FAST_XCD=1; do env $kv timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'knob': '$kv', 'kernel_ms': d['roofline']['kernel_ms'], 'frac': d['roofline']['frac']}))" >> $O/slow_knobs.jsonl; done
cat $O/slow_knobs.jsonl
timeout 200 python tools/bench_configs.py --configs 7 --layouts aos 2>/dev/null | grep "^{" | cut -c1-200 > $O/slow_configs7.jsonl; cat $O/slow_configs7.jsonl
cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/slow_sq -- python $R/bench.py --steps 10 --warmup 3 --no-cpu > /dev/null 2> $O/slow_sq.err
timeout 200 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_REQUEST --output-format csv -d $O/slow_tc -- python $R/bench.py --steps 10 --warmup 3 --no-cpu > /dev/null 2> $O/slow_tc.err
timeout 200 rocprofv3 --pmc TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ TCC_BUSY TCC_REQ --output-format csv -d $O/slow_tcc -- python $R/bench.py --steps 10 --warmup 3 --no-cpu > /dev/null 2> $O/slow_tcc.err
cd $R
python tools/pmc_summary.py $O/slow_sq $O/slow_tc $O/slow_tcc > $O/slow_pmc_summary.txt 2>&1; cat $O/slow_pmc_summary.txt | cut -c1-900; tail -2 $O/slow_tc.err $O/slow_tcc.err
find $O -name "*counter_collection.csv" -size +1M -delete
if python -c "import sys; sys.exit(0 if float('$MS') > 5.9 else 1)"; then
  echo "SLOW BOX: probing"
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu --layout soa > $O/slow_bench_soa.json 2>/dev/null
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu --tracks 250000 > $O/slow_bench_quarter.json 2>/dev/null
  timeout 300 python tools/bench_configs.py --configs 3a --layouts soa,aos 2>/dev/null | grep "^{" > $O/slow_configs.jsonl
  timeout 120 python tools/bench_resample.py --shapes 125x8000000,1000x8000 --iters 10 2>/dev/null | grep "^{" > $O/slow_resample.jsonl
  python - <<PY
import json
for f in ("slow_bench_soa", "slow_bench_quarter"):
    try:
        d = json.load(open("$O/" + f + ".json")); print(f, d["roofline"]["kernel_ms"], round(d["roofline"]["frac"], 3), d["under_load"].get("sclk_mhz"), d["under_load"].get("power_w"))
    except Exception as e:
        print(f, "failed", e)
for l in open("$O/slow_configs.jsonl"):
    r = json.loads(l); print(r["kernel"], round(r["ms"], 3), round(r["frac_of_8TBs"], 3))
print(open("$O/slow_resample.jsonl").read())
PY
fi
