#!/usr/bin/env python3
"""Does the relative placement of u and F matter for K1?  One big allocation; F's base is slid by `skew` bytes against a fixed u:
tools/k1_skew.py [n].  Uses the raw C entry point on pointer offsets (same kernel, same data)."""
import ctypes as C, os, sys, statistics
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
from ins_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
setup = ins.Setup(x=(np.linspace(0, 1, n + 1),) * 3, Re=1000.0)
N = n + 2
nvec = 3 * N**3
pad = 8 * 2**20 // 8  # 8 MiB of slack
pool = torch.zeros(2 * nvec + 2 * pad, dtype=torch.float64, device=setup.device)
base = pool.data_ptr()
base = (base + 2**21 - 1) // 2**21 * 2**21  # 2 MiB aligned
u_ptr = base
torch.manual_seed(0)
uview = pool[(u_ptr - pool.data_ptr()) // 8 : (u_ptr - pool.data_ptr()) // 8 + nvec]
uview.copy_(torch.randn(nvec, dtype=torch.float64, device=setup.device))
f0 = (u_ptr + 8 * nvec + 2**21 - 1) // 2**21 * 2**21
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def run(fptr, reps=5):
    _lib.call("ins_momentum_f64", setup.handle, 1e-3, C.c_void_p(u_ptr), C.c_void_p(fptr), setup.stream)
    e0.record()
    for _ in range(reps): _lib.call("ins_momentum_f64", setup.handle, 1e-3, C.c_void_p(u_ptr), C.c_void_p(fptr), setup.stream)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
skews = [0, 256, 1024, 4096, 8192, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, 3 << 19]
res = {s: [] for s in skews}
for rep in range(4):
    for s in skews:
        res[s].append(run(f0 + s))
print(f"u at 2MiB-aligned {hex(u_ptr)}, F at next 2MiB boundary + skew; component stride {8*N**3} B = {8*N**3/2**21:.3f} x 2MiB")
for s in skews:
    print(f"n={n} skew {s:8d} B: best {min(res[s]):.4f} ms median {statistics.median(res[s]):.4f} ms  {48.0*n**3/min(res[s])/1e6:.0f} GB/s", flush=True)
