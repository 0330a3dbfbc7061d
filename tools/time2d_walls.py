#!/usr/bin/env python3
"""ms/step of 2-D wall-bounded boxes (lid-driven cavity: cosine x cosine, Dirichlet; psolver_direct), RK44: tools/time2d_walls.py n [steps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
n = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
D = ins.DirichletBC
setup = ins.Setup(x=(ins.cosine_grid(0.0, 1.0, n),) * 2, Re=1000.0, boundary_conditions=((D(), D()), (D(), D((1.0, 0.0)))))
ps = ins.psolver_direct(setup)
u = ins.velocityfield(setup, lambda a, x, y: 0 * (x + y), 0.0, psolver=ps, doproject=False)
m = ins.RKMethods.RK44(); cache = ins.ode_method_cache(m, setup, ps)
st = ins.create_stepper(m, setup=setup, psolver=ps, u=u, t=0.0)
h = 0.05 * ins.get_cfl_timestep_(None, st.u, setup)
st = ins.timesteps_(m, st, h, 3, cache=cache); torch.cuda.synchronize()
t0 = time.perf_counter(); st = ins.timesteps_(m, st, h, steps, cache=cache); torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
print(f"2-D cavity {n}^2: {ms:.4f} ms/step = {n*n/ms/1e3:.1f} M cell-updates/s ({ms*1e6/(n*n):.3f} ns/cell), finite {bool(torch.isfinite(st.u).all())}", flush=True)
