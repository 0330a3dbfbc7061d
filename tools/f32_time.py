#!/usr/bin/env python3
"""ms/step of the fp32 family (RK44 + spectral projection): tools/f32_time.py n"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
f32 = ins.f32
n = int(sys.argv[1])
sp = ins.Setup(x=(np.linspace(0.0, 1.0, n + 1),) * 3, Re=1000.0)
u = f32.vectorfield32(sp); u.copy_(0.01 * torch.randn(u.shape, dtype=torch.float32, device=u.device)); f32.apply_bc_u32_(u, sp)
ps = f32.psolver_spectral32(sp)
m = ins.RKMethods.RK44()
cache = f32.ERKCache32(m, sp, ps)
p = f32.scalarfield32(sp); f32.project32_(u, sp, ps, p)
for _ in range(3): f32.timestep32_(cache, u, 1e-4)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 10
for _ in range(K): f32.timestep32_(cache, u, 1e-4)
torch.cuda.synchronize()
print(f"f32 n={n} opts={ {k: v for k, v in os.environ.items() if k.startswith('INS_')} }: {(time.perf_counter() - t0) * 1e3 / K:.3f} ms/step", flush=True)
