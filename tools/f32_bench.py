#!/usr/bin/env python3
"""fp32 family timings: K1 alone (24 B/cell) and one RK44 step, next to the fp64 kernels on the same box: tools/f32_bench.py N"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
f32 = ins.f32
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sp = ins.Setup(x=(np.linspace(0.0, 1.0, n + 1),) * 3, Re=1000.0)
u = f32.vectorfield32(sp); u.copy_(torch.randn(u.shape, dtype=torch.float32, device=u.device)); f32.apply_bc_u32_(u, sp)
F = f32.vectorfield32(sp)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(fn, reps=10):
    fn(); best = 1e9
    for _ in range(4):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / reps)
    return best
ms = t(lambda: f32.momentum32_(F, u, sp))
print(f"n={n} K1 fp32: {ms:.4f} ms = {24.0 * n**3 / ms / 1e6:.0f} GB/s ({24.0 * n**3 / ms / 1e6 / 8000:.3f} of 8 TB/s)", flush=True)
u64 = ins.vectorfield(sp); u64.copy_(u); F64 = ins.vectorfield(sp)
ms = t(lambda: ins.momentum_(F64, u64, None, 0.0, sp))
print(f"n={n} K1 fp64: {ms:.4f} ms = {48.0 * n**3 / ms / 1e6:.0f} GB/s", flush=True)
del u64, F64, F
ps = f32.psolver_spectral32(sp)
m = ins.RKMethods.RK44()
cache = f32.ERKCache32(m, sp, ps)
u.mul_(0.01); p = f32.scalarfield32(sp); f32.project32_(u, sp, ps, p)
ms = t(lambda: f32.timestep32_(cache, u, 1e-4), reps=5)
print(f"n={n} RK44 step fp32 (float stage kernels with in-register correction, the five solver passes on float2 spectra; INS_F32_FP64_SPECTRA=1: on the fp64 passes): {ms:.3f} ms = {n**3 / ms / 1e3:.0f} M cell-updates/s", flush=True)
ms = t(lambda: f32.timesteps32_(cache, u, 1e-4, 5), reps=1) / 5
print(f"n={n} RK44 step fp32, chained (timesteps32_, 5 steps per call): {ms:.3f} ms = {n**3 / ms / 1e3:.0f} M cell-updates/s", flush=True)
