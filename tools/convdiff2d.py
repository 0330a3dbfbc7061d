#!/usr/bin/env python3
"""A few launches of the generic 2-D momentum kernel at n^2 (for rocprofv3 --pmc): tools/convdiff2d.py n"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
n = int(sys.argv[1])
setup = ins.Setup(x=(np.linspace(0, 1, n + 1),) * 2)
u = ins.vectorfield(setup); u.copy_(torch.randn(u.shape, dtype=torch.float64, device=u.device)); F = ins.vectorfield(setup)
for _ in range(4): ins.momentum_(F, u, None, 0.0, setup)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ins.momentum_(F, u, None, 0.0, setup)
e1.record(); torch.cuda.synchronize()
print(f"momentum 2-D {n}^2: {e0.elapsed_time(e1)/10:.4f} ms", flush=True)
