#!/usr/bin/env python3
"""Profile target: RK44 steps of the fp32 family: tools/f32_prof.py n steps"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
f32 = ins.f32
n = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sp = ins.Setup(x=(np.linspace(0.0, 1.0, n + 1),) * 3, Re=1000.0)
u = f32.vectorfield32(sp); u.copy_(0.01 * torch.randn(u.shape, dtype=torch.float32, device=u.device)); f32.apply_bc_u32_(u, sp)
ps = f32.psolver_spectral32(sp)
m = ins.RKMethods.RK44()
cache = f32.ERKCache32(m, sp, ps)
p = f32.scalarfield32(sp); f32.project32_(u, sp, ps, p)
for _ in range(steps): f32.timestep32_(cache, u, 1e-4)
torch.cuda.synchronize(); print("done")
