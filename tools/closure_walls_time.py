#!/usr/bin/env python3
"""ms/step of wall-bounded boxes with the Smagorinsky closure through the native extended stage loop (tiled path): all-walls uniform 256³ and the
channel 256 x 128 x 256 (periodic x, z; tanh walls in y): tools/closure_walls_time.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
D, P = ins.DirichletBC, ins.PeriodicBC
def tanh_grid(a, b, n, g=1.5):
    s = np.linspace(-1, 1, n + 1)
    return a + (b - a) * (1 + np.tanh(g * s) / np.tanh(g)) / 2
cases = {
    "walls 256^3": (tuple(np.linspace(0, 1, 257) for _ in range(3)), ((D(), D()),) * 3),
    "channel 256x128x256": ((np.linspace(0, 4, 257), tanh_grid(0, 1, 128), np.linspace(0, 2, 257)), ((P(), P()), (D(), D()), (P(), P()))),
}
for name, (x, bc) in cases.items():
    for closure in (False, True):
        setup = ins.Setup(x=x, Re=1000.0, boundary_conditions=bc)
        if closure: setup.closure_model = ins.smagorinsky_closure(setup)
        ps = ins.psolver_direct(setup)
        u = ins.velocityfield(setup, lambda a, x, y, z: (a == 0) * np.sin(np.pi * y) * (1 + 0.1 * np.sin(2 * np.pi * z)) + 0 * x, 0.0, psolver=ps)
        m = ins.RKMethods.RK44(); cache = ins.ode_method_cache(m, setup, ps)
        st = ins.create_stepper(m, setup=setup, psolver=ps, u=u, t=0.0)
        th = 0.17 if closure else None
        for _ in range(2): st = ins.timestep_(m, st, 1e-4, cache=cache, θ=th)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): st = ins.timestep_(m, st, 1e-4, cache=cache, θ=th)
        torch.cuda.synchronize()
        print(f"{name} closure={'smagorinsky' if closure else 'no'}: {(time.perf_counter() - t0) * 200:.3f} ms/step", flush=True)
