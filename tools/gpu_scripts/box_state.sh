#!/bin/bash
# Everything about the leased box that could explain a 15 % spread of the headline kernel between boxes with identical
# clock readings: partition modes, perf level, power cap, RAS / ECC state, THP, XNACK, firmware.
#   bash tools/gpu_scripts/box_state.sh > gpurun_out/<dir>/box_state.txt
echo "== date"; date -u
echo "== uname"; uname -a
echo "== rocm-smi partitions / perf / power"; rocm-smi --showmemorypartition --showcomputepartition --showperflevel --showmaxpower --showpower --showclocks --showtemp 2>&1
echo "== rocm-smi ras / ecc"; rocm-smi --showrasinfo all 2>&1 | head -60
echo "== rocm-smi fw / vbios / serial-free ids"; rocm-smi --showvbios --showfwinfo 2>&1 | head -60
echo "== rocm-smi retired pages"; rocm-smi --showpids --showretiredpages 2>&1 | head -30
echo "== amd-smi static (if present)"; (command -v amd-smi > /dev/null && amd-smi static -g 0 2>&1 | head -120) || echo "no amd-smi"
echo "== amd-smi metric (if present)"; (command -v amd-smi > /dev/null && amd-smi metric -g 0 2>&1 | head -150) || true
echo "== THP"; cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag 2>&1
echo "== XNACK / HSA env"; env | grep -E "HSA_|XNACK|ROC|HIP" | sort
echo "== rocminfo (gfx / xnack / cu / clocks)"; rocminfo 2>&1 | grep -E "Name:|Compute Unit|Max Clock|xnack|Wavefront|Cacheline|L2|L3|Memory Properties" | head -60
echo "== cpu"; lscpu 2>&1 | grep -E "Model name|Socket|Core|Thread|^CPU\(s\)|NUMA|MHz" ; cat /sys/fs/cgroup/cpu.max 2>&1
echo "== numa / memory"; numactl -H 2>&1 | head -20; free -g 2>&1 | head -3
echo "== pcie link of gpu0"; for d in /sys/class/drm/card*/device; do [ -f $d/current_link_speed ] && echo "$d $(cat $d/current_link_speed) x$(cat $d/current_link_width) $(cat $d/vendor 2>/dev/null)"; done 2>&1 | head
echo "== amdgpu sysfs (pp / power)"; for d in /sys/class/drm/card*/device; do for f in power_dpm_force_performance_level pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk current_compute_partition current_memory_partition mem_info_vram_total; do [ -f $d/$f ] && { echo "-- $d/$f"; cat $d/$f; }; done; done 2>&1 | head -120
