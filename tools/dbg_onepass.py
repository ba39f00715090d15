"""Debug: reproduce one exp_onepass case and print the context of every differing slot."""
import os, sys, json
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from filterpy_amd import _engine as E
from oracle import resample_oracle as ro
from exp_onepass import weights

shape, strat, kind = sys.argv[1], int(sys.argv[2]), sys.argv[3]
Fn, Np = (int(v) for v in shape.split("x"))
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(11 + strat)
w = weights(kind, Fn, Np, dev, g)
u = torch.rand((Fn, Np) if strat else (Fn,), generator=g, device=dev, dtype=torch.float64)
new = torch.full((Fn, Np), -7, dtype=torch.int32, device=dev)
st = torch.zeros(Fn, dtype=torch.int32, device=dev)
os.environ["FK_RESAMPLE_PATH"] = "onepass"
import ctypes
lib = ctypes.CDLL(os.path.join(ROOT, "filterpy_amd", "csrc", "exp_build", "libop_phase.so"))
lib.fk_resample_workspace_bytes.restype = ctypes.c_size_t
nbytes = lib.fk_resample_workspace_bytes(ctypes.c_int64(Fn), ctypes.c_int64(Np))
ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=dev)
dcs = torch.zeros((Fn, Np), dtype=torch.float64, device=dev); dn = torch.zeros((Fn, Np), dtype=torch.int32, device=dev)
p = ctypes.c_void_p
assert lib.fk_debug_set_dump(p(dcs.data_ptr()), p(dn.data_ptr())) == 0
fn = lib.fk_resample_stratified_f64 if strat else lib.fk_resample_systematic_f64
rc = fn(ctypes.c_int64(Fn), ctypes.c_int64(Np), p(w.data_ptr()), p(u.data_ptr()), p(new.data_ptr()), p(st.data_ptr()),
        p(ws.data_ptr()), ctypes.c_size_t(nbytes), p(0))
assert rc == 0
torch.cuda.synchronize()
for f in range(Fn):
    wf = w[f].cpu().numpy(); uf = u[f].cpu().numpy() if strat else np.array([float(u[f])])
    ref, over = (ro.stratified_c if strat else ro.systematic_c)(wf, uf)
    got = new[f].cpu().numpy()
    bad = np.flatnonzero(got != ref)
    if len(bad) == 0:
        continue
    cs = np.cumsum(wf)
    gcs, gn = dcs[f].cpu().numpy(), dn[f].cpu().numpy()
    dcs_bad = np.flatnonzero(gcs != cs)
    print(json.dumps({"filter": f, "gpu_cs_differs_at": dcs_bad[:10].tolist(), "count": int(len(dcs_bad))}))
    for i in bad[:5]:
        ui = uf[i] if strat else uf[0]
        pos = (ui + i) / Np
        j0, j1 = int(ref[i]), int(got[i])
        lo = max(0, min(j0, j1) - 2); hi = max(j0, j1) + 3
        print(json.dumps({"filter": f, "slot": int(i), "ref": j0, "got": j1, "pos": pos.hex(), "u": float(ui).hex(),
                          "chunk_of_ref": j0 // 2048, "chunk_of_got": j1 // 2048, "j_in_chunk": j0 % 2048,
                          "cs": {int(j): cs[j].hex() for j in range(lo, min(hi, Np))},
                          "gpu_cs": {int(j): float(gcs[j]).hex() for j in range(lo, min(hi, Np))},
                          "gpu_n": {int(j): int(gn[j]) for j in range(lo, min(hi, Np))},
                          "w": {int(j): float(wf[j]).hex() for j in range(lo, min(hi, Np))},
                          "got_nbrs": got[max(0, i - 3): i + 4].tolist(), "ref_nbrs": ref[max(0, i - 3): i + 4].tolist()}))
print("done")
