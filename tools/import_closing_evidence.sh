#!/bin/bash
# Copies what the closing lease (tools/gpu_scripts/r03_final.sh -> gpurun_out/r03fin) produced into profiles/r03 under the names
# DESIGN.md and profiles/README.md cite, reduces the PMC passes into profiles/pmc_traffic.json and regenerates DESIGN's tables.
#   bash tools/import_closing_evidence.sh [gpurun_out/r03fin] [profiles/r03]   (regenerates all three generated blocks of DESIGN.md)
set -e
S=${1:-gpurun_out/r03fin}
D=${2:-profiles/r03}
cp $S/bench_default.json $S/bench_default_placement_none.json $S/bench_under_rocprof_stats.json $S/bench_under_rocprof_pmc_fetch.json \
   $S/bench_under_rocprof_pmc_write.json $S/configs_all.jsonl $S/resample_shapes.jsonl $S/kernel_durations.txt \
   $S/kernel_durations_bench_last20.txt $S/bench_kernel_trace_fk.csv $S/pytest_gpu_full.log $S/box_state.txt $S/smoke.log \
   $S/bench_c5_1000x8000.json $S/bench_c5_125x8000.json $S/bench_c5_125x8000000.json $D/
cp $(ls -t $S/prof_stats/runc/*_kernel_stats.csv | head -1) $D/kf_c2_aos_kernel_stats.csv
cp $(ls -t $S/prof_cfg/runc/*_kernel_stats.csv | head -1) $D/configs_all_kernel_stats.csv
cp $(ls -t $S/rs_stats/runc/*_kernel_stats.csv | head -1) $D/resample_kernel_stats.csv
cp $S/prof_fetch_fk.csv $D/kf_c2_aos_pmc_fetch.csv
cp $S/prof_write_fk.csv $D/kf_c2_aos_pmc_write.csv
head -1 $S/bench_force_dist.json > $D/bench_force_dist_1rank_nccl.json
sed -n '2,$p' $S/bench_force_dist.json > $D/bench_force_dist_rccl_banner.txt
head -1 $S/bench_c5_force_dist.json > $D/bench_c5_force_dist_1rank_nccl.json
cp $S/ukf_kernels.jsonl $D/ukf_kernels_closing.jsonl
python - "$S" <<'PY'
import json, sys
p = "profiles/pmc_traffic.json"
t = json.load(open(p))
h = json.load(open(sys.argv[1] + "/pmc_headline.json"))
t["aos"].update(FETCH_SIZE_KiB=h["FETCH_SIZE_KiB"], WRITE_SIZE_KiB=h["WRITE_SIZE_KiB"], hbm_bytes_per_launch=h["hbm_bytes_per_launch"], round="r03",
                source="profiles/r03/kf_c2_aos_pmc_fetch.csv + kf_c2_aos_pmc_write.csv (mean of %d launches each of `python bench.py --steps 20 --warmup 5 --no-cpu` under rocprofv3 --pmc, separate passes; FETCH_SIZE doubled per the gfx950 correction; tools/pmc_reduce.py)" % h["launches"][0])
json.dump(t, open(p, "w"), indent=2)
open(p, "a").write("\n")
print("pmc:", t["aos"]["hbm_bytes_per_launch"], "bytes per launch")
PY
python tools/make_design_table.py $D --write
tail -1 $D/pytest_gpu_full.log
