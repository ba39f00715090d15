#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03u
mkdir -p $O
cd $R
for d in "7 1" "9 4" "8 4"; do set -- $d; NX=$1 NZ=$2 timeout 200 python tools/debug/ex_aos_mask.py > $O/ex_aos_mask_$1_$2.log 2>&1; echo "== $1 $2"; cut -c1-300 $O/ex_aos_mask_$1_$2.log | tail -42; done
