#!/bin/bash
# Copies what the closing lease (tools/gpu_scripts/r06_final.sh -> gpurun_out/r06fin) produced into profiles/r06 under the names
# DESIGN.md / docs/MEASUREMENTS.md cite, reduces the PMC passes into profiles/pmc_traffic.json and regenerates the tables.
#   bash tools/import_closing_evidence.sh [gpurun_out/r06fin] [profiles/r06]
set -e
S=${1:-gpurun_out/r06fin}
D=${2:-profiles/r06}
mkdir -p $D/c5
cp $S/bench_default.json $S/bench_placement_probe.json $S/bench_placement_none.json $S/bench_under_rocprof_stats.json \
   $S/bench_under_rocprof_pmc_fetch.json $S/bench_under_rocprof_pmc_write.json $S/configs_all.jsonl $S/configs_all_kernel_durations.txt \
   $S/configs_all_kernel_stats.csv $S/resample_shapes.jsonl $S/kernel_durations.txt $S/kernel_durations_bench_last20.txt \
   $S/bench_kernel_trace_fk.csv $S/pytest_gpu_full.log $S/box_state.txt $S/smoke.log \
   $S/ukf_kernels.jsonl $S/ukf_kernels_index_order.jsonl $S/ukf_kernel_durations.txt $S/ukf_kernel_durations_index_order.txt $D/
cp $S/configs_traffic.jsonl $S/imm_outputs.jsonl $D/ 2>/dev/null || true
cp $S/prof_fetch_fk.csv $D/kf_c2_aos_pmc_fetch.csv
cp $S/prof_write_fk.csv $D/kf_c2_aos_pmc_write.csv
cp $S/kernel_durations_round3_onepass.txt $S/resample_under_stats_round3_onepass.jsonl $S/onepass_pmc.json $S/onepass_phase_clocks.jsonl $S/bench_strong_1rank.json $S/bench_api.jsonl $S/c_abi_multi_gpu.log $D/ 2>/dev/null || true
cp $S/bench_c5_1000x8000.json $S/bench_c5_125x8000.json $S/bench_c5_125x8000000.json $S/bench_c5_force_dist_1rank_nccl.json $D/c5/
cp $S/bench_force_dist_1rank_nccl.json $D/bench_force_dist_1rank_nccl.json
python - "$D" <<'PY'
# HBM traffic per launch of the kernel the DEFAULT placement runs (two placed arrays: the plain kf_fast instantiation) from the
# two separate PMC passes of `python bench.py --steps 20 --warmup 5 --no-cpu` (rows of every fk:: kernel are in the CSVs; the
# interleaved instantiation's entry, aos_interleave, comes from the lease that ran it as the timed kernel: profiles/r06/kf_c2_aos_il_*)
import csv, json, sys
d = sys.argv[1]
plain = "kf_fast_kernel<4, 2, 0, false, true, false, 0, false, false, false, false>"
def mean(path, counter):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and plain in r["Kernel_Name"]]
    return sum(v) / len(v), len(v)
f, nf = mean(d + "/kf_c2_aos_pmc_fetch.csv", "FETCH_SIZE")
w, nw = mean(d + "/kf_c2_aos_pmc_write.csv", "WRITE_SIZE")
p = "profiles/pmc_traffic.json"
t = json.load(open(p))
t["aos"] = dict(FETCH_SIZE_KiB=f, WRITE_SIZE_KiB=w, hbm_bytes_per_launch=int(round((2 * f + w) * 1024)), round="r06",
                fetch_csv=d + "/kf_c2_aos_pmc_fetch.csv", write_csv=d + "/kf_c2_aos_pmc_write.csv",
                source="%s/kf_c2_aos_pmc_fetch.csv + kf_c2_aos_pmc_write.csv (mean of %d / %d launches of the plain kf_fast<4,2,aos> kernel -- the one the default placement runs -- under `python bench.py --steps 20 --warmup 5 --no-cpu`, FETCH_SIZE doubled per the gfx950 correction)" % (d, nf, nw))
json.dump(t, open(p, "w"), indent=2)
open(p, "a").write("\n")
json.dump(dict(kernel=plain, FETCH_SIZE_KiB=f, WRITE_SIZE_KiB=w, launches=[nf, nw], hbm_bytes_per_launch=t["aos"]["hbm_bytes_per_launch"]), open(d + "/pmc_headline.json", "w"))
print("pmc:", t["aos"]["hbm_bytes_per_launch"], "bytes per launch")
PY
python tools/make_design_table.py $D --write
tail -1 $D/pytest_gpu_full.log
