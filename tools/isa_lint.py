#!/usr/bin/env python3
"""Static facts of the built gfx950 kernels, no GPU needed: code size (the instruction cache holds 64 KB), registers, scratch,
LDS, waterfall loops (a buffer op whose descriptor the compiler could not prove uniform) and exec-masked regions inside the time
loop that contain a cross-lane operation (DPP / ds_swizzle / ds_bpermute under a divergent branch would read disabled lanes).
A heuristic over the disassembly in LAYOUT order (a region = s_*_saveexec up to the next scalar write of exec): good for spotting,
not a proof -- a block the compiler placed out of line (the odd-tail branch of a copy-out, say) makes the linear scan span code
that is not under that mask; look at what it flags.

    python tools/isa_lint.py filterpy_amd/csrc/build/ukf_mlg_16.o [...]        # objects built by csrc/Makefile
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def device_elf(obj, tmp):
    out = os.path.join(tmp, os.path.basename(obj) + ".elf")
    # host objects carry the device code object as an offload bundle in a section; extract it next to a copy
    cp = os.path.join(tmp, os.path.basename(obj))
    subprocess.check_call(["cp", obj, cp])
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", cp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    cand = [f for f in os.listdir(tmp) if f.startswith(os.path.basename(obj)) and TARGET in f]
    if not cand:
        raise RuntimeError(f"no {TARGET} bundle in {obj}")
    os.replace(os.path.join(tmp, cand[0]), out)
    return out


def kernels(elf):
    notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", elf], text=True)
    info = {}
    for blk in notes.split("- .agpr_count")[1:]:
        def grab(key):
            m = re.search(r"\." + key + r":\s+(\S+)", blk)
            return m.group(1) if m else "?"
        name = grab("name")
        info[name] = dict(lds=int(grab("group_segment_fixed_size")), scratch=int(grab("private_segment_fixed_size")), vgpr=int(grab("vgpr_count")))
    syms = subprocess.check_output([f"{LLVM}/llvm-readelf", "-sW", elf], text=True)
    for l in syms.splitlines():
        f = l.split()
        if len(f) >= 8 and f[3] == "FUNC" and f[7] in info:
            info[f[7]]["code"] = int(f[2])
    return info


def lint_asm(elf):
    dis = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", elf], text=True)
    res, cur, lines = {}, None, []
    def flush():
        if cur is None:
            return
        wf = sum(1 for i, l in enumerate(lines) if "s_and_saveexec" in l and any("v_readfirstlane" in p for p in lines[max(0, i - 8):i])
                 and any("v_cmp_eq" in p for p in lines[max(0, i - 8):i]))
        # the time loop: the longest backward branch.  llvm-objdump: "<op> ... // <address>: <encoding> <symbol+0xOFFSET>"
        addr = []
        for l in lines:
            m = re.search(r"//\s*([0-9A-Fa-f]+):", l)
            addr.append(int(m.group(1), 16) if m else None)
        known = [(a, i) for i, a in enumerate(addr) if a is not None]
        base = known[0][0] if known else 0
        span = (0, 0)
        for i, l in enumerate(lines):
            m = re.search(r"s_cbranch_scc[01]\s.*<\S+?\+0x([0-9a-f]+)>", l)      # (uniform loops close on scc; layout jumps use s_branch / exec)
            if m and addr[i] is not None:
                tgt = base + int(m.group(1), 16)
                if tgt < addr[i]:
                    j = next((k for a, k in known if a >= tgt), i)
                    if i - j > span[1] - span[0]:
                        span = (j, i)
        bad_in, bad_out, depth_start = 0, 0, None
        for i, l in enumerate(lines):
            if "saveexec" in l:
                depth_start = i
            elif depth_start is not None and re.search(r"\bs_\w+\s+exec\b", l):           # any scalar write of exec closes the region
                if any(("dpp" in p or "ds_swizzle" in p or "ds_bpermute" in p) for p in lines[depth_start:i]):
                    if span[0] <= depth_start <= span[1]:
                        bad_in += 1
                    else:
                        bad_out += 1
                depth_start = None
        res[cur] = dict(waterfalls=wf, masked_crosslane=bad_in, masked_outside=bad_out, loop=span[1] - span[0])
    for l in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
        if m:
            flush()
            cur, lines = m.group(1), []
        elif cur is not None:
            lines.append(l)
    flush()
    return res


def dma_wait_margins(elf):
    """LDS-DMA prefetches (`buffer_load_dword ... lds`) in a time loop are waited for by a hand-counted `s_waitcnt vmcnt(K)` at
    the top of the NEXT trip: K must not exceed the vector-memory instructions a trip issues BEHIND the request (vmcnt retires in
    order: with fewer than K younger operations the wait returns while the DMA is still in flight and the kernel reads a stale
    image -- ADVICE r4: the count is a lower bound written in the source, the compiler may merge adjacent stores).
    Returns {kernel: (K, younger, checkable)} for every kernel with such a loop: K = the largest explicit vmcnt(>= 1) that sits
    between the loop head and the first DMA request of the loop, younger = VMEM instructions from behind the loop's LAST DMA
    request to the loop's end plus from its head to that wait, counted in layout order -- which is program order only where the
    time loop holds no inner loop (checkable; IMM banks, kf_fast): the several-lane kernels copy their tiles out in rolled loops
    whose stores this static count sees once, not once per trip."""
    dis = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", elf], text=True)
    out, cur, lines = {}, None, []

    def flush():
        if cur is None:
            return
        addr = []
        for l in lines:
            m = re.search(r"//\s*([0-9A-Fa-f]+):", l)
            addr.append(int(m.group(1), 16) if m else None)
        known = [(a, i) for i, a in enumerate(addr) if a is not None]
        if not known:
            return
        base = known[0][0]
        vmem = lambda l: re.match(r"\s*(buffer|global|flat|scratch)_(load|store|atomic)", l) is not None      # noqa: E731
        for i, l in enumerate(lines):
            m = re.search(r"s_cbranch_(scc[01]|vccn?z|execn?z)\s.*<\S+?\+0x([0-9a-f]+)>", l)
            if not (m and addr[i] is not None):
                continue
            tgt = base + int(m.group(2), 16)
            if tgt >= addr[i]:
                continue
            j = next((k for a, k in known if a >= tgt), i)
            body = lines[j:i + 1]
            dma = [k for k, b in enumerate(body) if re.match(r"\s*buffer_load_\w+ .*\blds\b", b)]
            if not dma:
                continue
            waits = [(k, int(mm.group(1))) for k, b in enumerate(body[:dma[0]]) for mm in [re.search(r"s_waitcnt vmcnt\((\d+)\)", b)] if mm and int(mm.group(1)) >= 1]
            if not waits:
                continue
            wk, K = max(waits, key=lambda t: t[1])
            younger = sum(1 for b in body[dma[-1] + 1:] if vmem(b)) + sum(1 for b in body[:wk] if vmem(b))
            inner = False
            for k, b in enumerate(body[:-1]):
                mm = re.search(r"s_cbranch_\w+\s.*<\S+?\+0x([0-9a-f]+)>", b)
                if mm and addr[j + k] is not None and base + int(mm.group(1), 16) < addr[j + k]:
                    inner = True
            prev = out.get(cur)
            if prev is None or K - younger > prev[0] - prev[1]:
                out[cur] = (K, younger, not inner)
    for l in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
        if m:
            flush()
            cur, lines = m.group(1), []
        elif cur is not None:
            lines.append(l)
    flush()
    return out


def short(name):
    m = re.search(r"\d+(ukf_mlg_rts_kernel|ukf_mlg_kernel|[a-z_0-9]+_kernel)I(.*?)EEv", name)
    return (m.group(1) + "<" + re.sub(r"L[ib](\d+)E", r"\1,", m.group(2)).rstrip(",") + ">") if m else name[:60]


if __name__ == "__main__":
    print("# masked x-lane: exec-masked regions INSIDE the time loop that contain a DPP / swizzle (outside it: the epilogue's status OR under")
    print("# `if (owner)`, where whole lane groups are on or off together -- counted in the last column)")
    print(f"{'kernel':58s} {'code':>7s} {'vgpr':>5s} {'scratch':>8s} {'lds':>7s} {'waterfall':>9s} {'masked x-lane':>13s} {'(outside)':>9s}")
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sys.argv[1:]:
            elf = device_elf(obj, tmp)
            info, lint = kernels(elf), lint_asm(elf)
            for name in sorted(info, key=short):
                k, q = info[name], lint.get(name, {})
                flag = "  <-- > 64 KB" if k.get("code", 0) > 65536 else ""
                print(f"{short(name):58s} {k.get('code', 0):7d} {k['vgpr']:5d} {k['scratch']:8d} {k['lds']:7d} {q.get('waterfalls', '?'):>9} {q.get('masked_crosslane', '?'):>13} {q.get('masked_outside', '?'):>9}{flag}")
