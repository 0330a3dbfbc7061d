#!/usr/bin/env python3
"""Run the momentum-RHS kernel a few times at one configuration (for rocprofv3 --pmc runs).
Usage: python tools/k1_one.py n rows zchunk [iters]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins

n, rows, zc = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
lib = ins._lib.load()
lib.ins_tune_fast3d.argtypes = [C.c_int, C.c_int]
lib.ins_tune_fast3d(rows, zc)
setup = ins.Setup(x=(np.linspace(0, 1, n + 1),) * 3, Re=1000.0)
torch.manual_seed(0)
u = ins.vectorfield(setup)
u.copy_(torch.randn(u.shape, dtype=torch.float64, device=u.device))
ins.apply_bc_u_(u, 0.0, setup)
F = ins.vectorfield(setup)
for _ in range(iters):
    ins.momentum_(F, u, None, 0.0, setup)
torch.cuda.synchronize()
print("done")
