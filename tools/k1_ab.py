#!/usr/bin/env python3
"""In-process A/B of K1 variants on ONE allocation (timings across processes differ by up to 10% with the placement of the
arrays): tools/k1_ab.py [n] — plain momentum kernel at n^3, every variant timed round-robin `reps` times, best and median."""
import os, sys, statistics
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
from ins_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lib = _lib.load()
tune = lib.ins_tune_flux64
setup = ins.Setup(x=(np.linspace(0, 1, n + 1),) * 3, Re=1000.0)
torch.manual_seed(0)
u = ins.vectorfield(setup); u.copy_(torch.randn(u.shape, dtype=torch.float64, device=u.device)); ins.apply_bc_u_(u, 0.0, setup)
F = ins.vectorfield(setup)
# (label, disable, rows, zc, skel, burst, lds)
variants = [("old62", 1, 0, 0, 0, 0, 0, 0)]
for xw in (2, 4):
    for r in (4,):
        for zc in (16, 32, 64):
            variants.append((f"f64 R{r} zc{zc} xw{xw}", 0, r, zc, 0, 0, 0, xw))
variants += [("skel R4 zc32 xw2", 0, 4, 32, 1, 0, 0, 2), ("f64 R4 zc32 xw2 lds70k", 0, 4, 32, 0, 0, 70000, 2)]
times = {v[0]: [] for v in variants}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(6):
    for (label, dis, r, zc, sk, bu, lds, xw) in variants:
        tune(dis, r, -1, zc, xw, sk, bu, lds)
        ins.momentum_(F, u, None, 0.0, setup)
        e0.record()
        for _ in range(5): ins.momentum_(F, u, None, 0.0, setup)
        e1.record(); torch.cuda.synchronize()
        times[label].append(e0.elapsed_time(e1) / 5)
for label, ts in times.items():
    print(f"n={n} {label:24s} best {min(ts):.4f} ms  median {statistics.median(ts):.4f} ms  {48.0*n**3/min(ts)/1e6:.0f} GB/s", flush=True)
