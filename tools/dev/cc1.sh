#!/bin/bash
# compile ONE unit of filterpy_amd/csrc for gfx950 into /tmp and print its kernels' code size / registers / scratch (tools/isa_lint.py)
#   tools/dev/cc1.sh resample_whole.hip [-ffp-contract=off -DFOO=1 ...]        (-S: also leaves /tmp/<unit>.s)
R=$(cd "$(dirname "$0")/../.." && pwd)
U=$1; shift
B=$(basename "$U" .hip)
cd "$R/filterpy_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function "$@" -c "$U" -o "/tmp/$B.o" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function "$@" -S --cuda-device-only "$U" -o "/tmp/$B.s" 2>/dev/null
python "$R/tools/isa_lint.py" "/tmp/$B.o" | grep -v "^#"
