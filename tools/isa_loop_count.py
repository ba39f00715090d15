#!/usr/bin/env python3
"""Static instruction mix of the hottest loop of every kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only):
    python tools/isa_loop_count.py file.s [substring of the kernel name]
For each kernel: the backward branch with the longest span is taken as the time loop; prints instructions, VALU
instructions, fp64 VALU instructions, FMAs, LDS / VMEM operations, scratch operations inside it.  A wave64 VALU
instruction occupies its SIMD for >= 4 clocks, so  VALU x 4 x (wave-steps per SIMD)  is a floor for the kernel's time --
the number DESIGN.md quotes as the fp64-VALU bound of the arithmetic-bound kernels."""
import re
import sys
from collections import Counter


def kernels(lines):
    cur, start = None, 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            if cur:
                yield cur, start, i
            cur, start = m.group(1), i
    if cur:
        yield cur, start, len(lines)


def main():
    lines = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, a, b in kernels(lines):
        if want not in name:
            continue
        seg = lines[a:b]
        labels = {m.group(1): i for i, l in enumerate(seg) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        best = None
        for i, l in enumerate(seg):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and labels.get(m.group(1), i) < i and (best is None or i - labels[m.group(1)] > best[1] - best[0]):
                best = (labels[m.group(1)], i)
        if not best:
            continue
        body = [l.split()[0] for l in seg[best[0]:best[1]] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        c = Counter(body)
        tot = lambda pred: sum(v for k, v in c.items() if pred(k))
        print(f"{name[:110]}\n   loop: {len(body)} instr, VALU {tot(lambda k: k.startswith('v_'))}, fp64 {tot(lambda k: 'f64' in k)}, "
              f"fma {tot(lambda k: 'fma' in k and 'f64' in k)}, LDS {tot(lambda k: k.startswith('ds_'))}, "
              f"VMEM {tot(lambda k: k.startswith(('buffer_', 'global_', 'flat_')))}, scratch {tot(lambda k: k.startswith('scratch_'))}, "
              f"DPP/readlane {tot(lambda k: 'dpp' in k or 'readlane' in k or 'writelane' in k)}")


if __name__ == "__main__":
    main()
