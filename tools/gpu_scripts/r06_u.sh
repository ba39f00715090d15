#!/bin/bash
# Round 6, lease u: why does the element-major (4,2) bank run at 0.55 where NumPy order runs at 0.8?  The same bench command in both
# record orders: N dependence, then three PMC passes each (never combined with a trace domain).
ulimit -c 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06u
mkdir -p $O
export TMPDIR=/tmp
cd $R
export FK_BENCH_SKIP_PROBE=1
for lay in aos soa; do
for n in 100000 300000 500000 1000000 2000000; do
  timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu --no-configs --placement none --layout $lay --tracks $n 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print(json.dumps({'layout': '$lay', 'tracks': $n, 'kernel_ms': r['kernel_ms'], 'frac': r['frac'], 'kernel': r['kernel']}))" >> $O/n_dependence.jsonl
done
done
cat $O/n_dependence.jsonl | cut -c1-160
cd /tmp
for lay in aos soa; do
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-configs --placement none --layout $lay"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_$lay -- $B > /dev/null 2> $O/pmc_sq_$lay.err
timeout 300 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_REQUEST --output-format csv -d $O/pmc_tcp_$lay -- $B > /dev/null 2> $O/pmc_tcp_$lay.err
timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ TCC_BUSY TCC_REQ --output-format csv -d $O/pmc_tcc_$lay -- $B > /dev/null 2> $O/pmc_tcc_$lay.err
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq2_$lay -- $B > /dev/null 2> $O/pmc_sq2_$lay.err
cd $R
python tools/pmc_summary.py $O/pmc_sq_$lay $O/pmc_tcp_$lay $O/pmc_tcc_$lay $O/pmc_sq2_$lay > $O/pmc_$lay.json; cut -c1-900 $O/pmc_$lay.json
cd /tmp
done
