#!/bin/bash
# Round 5: the whole GPU suite on the library as shipped (per-shape wave targets, (9,x) on four lanes, extras on the masked twin)
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05s
mkdir -p $O
cd $R
timeout 330 python -m pytest tests -m gpu -q -p no:cacheprovider -rs 2>&1 | tail -8 | cut -c1-300 | tee $O/pytest_gpu_tail.txt
