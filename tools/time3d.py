#!/usr/bin/env python3
"""ms/step of the 3-D periodic path at n^3 (any n: power-of-two sizes run the own FFT passes, others rocFFT): tools/time3d.py n"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
n = int(sys.argv[1]); steps = 10
L = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
setup = ins.Setup(x=(np.linspace(0, L, n + 1),) * 3, Re=1000.0)
ps = ins.psolver_spectral(setup)
u = ins.random_field(setup, kp=4, psolver=ps, seed=0)
m = ins.RKMethods.RK44(); cache = ins.ode_method_cache(m, setup, ps)
st = ins.create_stepper(m, setup=setup, psolver=ps, u=u, t=0.0)
st = ins.timesteps_(m, st, 1e-4, 3, cache=cache); torch.cuda.synchronize()
t0 = time.perf_counter(); st = ins.timesteps_(m, st, 1e-4, steps, cache=cache); torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
print(f"3-D {n}^3 L={L:.4f}: {ms:.3f} ms/step = {n**3/ms/1e3:.0f} M cell-updates/s ({ms*1e6/n**3:.3f} ns/cell)", flush=True)
