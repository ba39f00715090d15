#!/bin/bash
# Copies what the closing lease (tools/gpu_scripts/r05_final.sh -> gpurun_out/r05fin) produced into profiles/r04 under the names
# DESIGN.md / docs/MEASUREMENTS.md cite, reduces the PMC passes into profiles/pmc_traffic.json and regenerates the tables.
#   bash tools/import_closing_evidence.sh [gpurun_out/r05fin] [profiles/r05]
set -e
S=${1:-gpurun_out/r05fin}
D=${2:-profiles/r05}
mkdir -p $D/c5
cp $S/bench_default.json $S/bench_placement_probe.json $S/bench_placement_none.json $S/bench_under_rocprof_stats.json \
   $S/bench_under_rocprof_pmc_fetch.json $S/bench_under_rocprof_pmc_write.json $S/configs_all.jsonl $S/configs_all_kernel_durations.txt \
   $S/configs_all_kernel_stats.csv $S/resample_shapes.jsonl $S/kernel_durations.txt $S/kernel_durations_bench_last20.txt \
   $S/bench_kernel_trace_fk.csv $S/pytest_gpu_full.log $S/box_state.txt $S/smoke.log $S/pmc_headline.json \
   $S/ukf_kernels.jsonl $S/ukf_kernels_index_order.jsonl $S/ukf_kernel_durations.txt $S/ukf_kernel_durations_index_order.txt $D/
cp $S/prof_fetch_fk.csv $D/kf_c2_aos_il_pmc_fetch.csv
cp $S/prof_write_fk.csv $D/kf_c2_aos_il_pmc_write.csv
cp $S/kernel_durations_round4_resampler.txt $S/bench_api.jsonl $S/c_abi_multi_gpu.log $D/ 2>/dev/null || true
cp $S/bench_c5_1000x8000.json $S/bench_c5_125x8000.json $S/bench_c5_125x8000000.json $S/bench_c5_force_dist_1rank_nccl.json $D/c5/
cp $S/bench_force_dist_1rank_nccl.json $D/bench_force_dist_1rank_nccl.json
python - "$S" <<'PY'
import json, sys
p = "profiles/pmc_traffic.json"
t = json.load(open(p))
h = json.load(open(sys.argv[1] + "/pmc_headline.json"))
t["aos_interleave"] = dict(FETCH_SIZE_KiB=h["FETCH_SIZE_KiB"], WRITE_SIZE_KiB=h["WRITE_SIZE_KiB"], hbm_bytes_per_launch=h["hbm_bytes_per_launch"], round="r05",
                           fetch_csv="profiles/r05/kf_c2_aos_il_pmc_fetch.csv", write_csv="profiles/r05/kf_c2_aos_il_pmc_write.csv",
                           source="profiles/r05/kf_c2_aos_il_pmc_fetch.csv + kf_c2_aos_il_pmc_write.csv (mean of %d / %d launches of the IL kernel under `python bench.py --steps 20 --warmup 5 --no-cpu`, the default placement; FETCH_SIZE doubled per the gfx950 correction; tools/pmc_reduce.py)" % tuple(h["launches"]))
json.dump(t, open(p, "w"), indent=2)
open(p, "a").write("\n")
print("pmc:", t["aos_interleave"]["hbm_bytes_per_launch"], "bytes per launch")
PY
python tools/make_design_table.py $D --write
tail -1 $D/pytest_gpu_full.log
