import ctypes, os, sys, torch, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
lib = ctypes.CDLL(os.path.join(ROOT, "filterpy_amd", "csrc", "exp_build", "libop_phase.so"))
u = float.fromhex("0x1.b3c009db9ee72p-1"); Np = 8000000
cs = [float.fromhex(x) for x in ["0x1.604196c382d02p-2", "0x1.6041a121e50d1p-2", "0x1.6041a121ee90fp-2", "0x1.6041a121eed2bp-2", "0x1.6041a124693f3p-2"]]
c = torch.tensor(cs, dtype=torch.float64, device="cuda")
out = torch.zeros(len(cs), dtype=torch.int32, device="cuda"); aux = torch.zeros(4 * len(cs), dtype=torch.float64, device="cuda")
p = ctypes.c_void_p
rc = lib.fk_debug_n_boundary(ctypes.c_int(len(cs)), p(c.data_ptr()), ctypes.c_int(Np), ctypes.c_double(u), p(out.data_ptr()), p(aux.data_ptr()))
print(rc, out.cpu().tolist(), "expected [2752001, 2752002, 2752003, 2752003, 2752003]")
print([float(x).hex() for x in aux.cpu().numpy()])
