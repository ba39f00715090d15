#!/usr/bin/env python3
"""Compile ONE instantiation of the fused UKF kernels (two seconds instead of the two minutes of the whole unit) and print what
the compiler made of it: registers, scratch, LDS, and the instruction mix of the time loop.

    python tools/ukf_one_kernel.py fwd 6 3 soa            # ukf_linear_kernel<6, 3, LAYOUT_SOA, true, true>
    python tools/ukf_one_kernel.py rts 6 aos --dma        # ukf_linear_rts_kernel<6, LAYOUT_AOS, true, true, true>
    python tools/ukf_one_kernel.py fwd 9 4 aos --padded   # the padded instantiation of the class

Uses csrc/ukf_kernels.hip's parts 91 / 92 (the kernel templates alone).  The listing stays in /tmp for a closer look."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "filterpy_amd", "csrc")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kind", choices=["fwd", "rts"])
    ap.add_argument("dims", nargs="+", help="fwd: NX NZ layout; rts: NX layout")
    ap.add_argument("--padded", action="store_true")
    ap.add_argument("--dma", action="store_true")
    ap.add_argument("--index-order", action="store_true", help="the index-order sums (PAIRED = false)")
    ap.add_argument("--sp", action="store_true", help="fwd, element-major: 16-byte stores of two element rows (SP = true)")
    ap.add_argument("--mllvm", action="append", default=[], help="extra -mllvm option (repeatable), e.g. --mllvm=-amdgpu-sched-strategy=max-ilp")
    a = ap.parse_args()
    lay = {"soa": "fk::LAYOUT_SOA", "aos": "fk::LAYOUT_AOS"}[a.dims[-1]]
    exact = "false" if a.padded else "true"
    paired = "false" if a.index_order else "true"
    if a.kind == "fwd":
        nx, nz = int(a.dims[0]), int(a.dims[1])
        inst = (f"template __global__ void fk::ukf_linear_kernel<{nx}, {nz}, {lay}, {exact}, {paired}, {'true' if a.sp else 'false'}>(const fk::UkfArgs, const double *, "
                "const double *, const double *, const double *, const double *, const double *, const double *, const uint8_t *);")
        part = 91
    else:
        nx = int(a.dims[0])
        inst = (f"template __global__ void fk::ukf_linear_rts_kernel<{nx}, {lay}, {exact}, {paired}, {'true' if a.dma else 'false'}>("
                "const fk::UkfRtsArgs, const double *, const double *, const double *, const double *);")
        part = 92
    src = f"/tmp/ukf_one_{os.getpid()}.hip"
    asm = src[:-4] + ".s"
    with open(src, "w") as fh:
        fh.write(f'#define FK_UKF_PART {part}\n#include "{os.path.join(CSRC, "ukf_kernels.hip")}"\n{inst}\n')
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-Wno-pass-failed", "-S",
           "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", src, "-o", asm]
    for opt in a.mllvm:
        cmd += ["-mllvm", opt]
    out = subprocess.run(cmd, capture_output=True, text=True)
    for line in out.stderr.splitlines():
        if "error" in line:
            print(line)
        for key in ("Function Name", " VGPRs:", "AGPRs:", "ScratchSize", "Occupancy", "LDS Size"):
            if key in line:
                print(line.split("remark:")[-1].split("[-R")[0].strip())
    if out.returncode:
        sys.exit(out.returncode)
    print(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_loop_count.py"), asm], capture_output=True, text=True).stdout.strip())
    print("listing:", asm)


if __name__ == "__main__":
    main()
