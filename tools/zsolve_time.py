#!/usr/bin/env python3
"""Time the spectral Poisson solve (5 own FFT passes) at n^3: tools/zsolve_time.py n  (knobs: INS_ZSOLVE_TK)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
n = int(sys.argv[1])
setup = ins.Setup(x=(np.linspace(0, 1, n + 1),) * 3, Re=1000.0)
ps = ins.psolver_spectral(setup)
p = ins.scalarfield(setup); p.copy_(torch.randn(p.shape, dtype=torch.float64, device=p.device))
for _ in range(3): ps(p)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(3):
    e0.record()
    for _ in range(10): ps(p)
    e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / 10)
print(f"n={n} TK={os.environ.get('INS_ZSOLVE_TK','-')}: poisson solve {best:.4f} ms", flush=True)
