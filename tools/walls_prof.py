#!/usr/bin/env python3
"""All-walls box (Dirichlet on six sides, lid on +y; uniform or cosine grid, psolver_direct) step loop for rocprofv3: tools/walls_prof.py n steps [cos]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
x = (ins.cosine_grid(0.0, 1.0, n), ins.cosine_grid(0.0, 1.0, n), ins.cosine_grid(0.0, 1.0, n)) if len(sys.argv) > 3 and sys.argv[3] == "cos" else (np.linspace(0, 1, n + 1),) * 3
D, P = ins.DirichletBC, ins.PeriodicBC
setup = ins.Setup(x=x, Re=1000.0, boundary_conditions=((D(), D()), (D(), D((1.0, 0.2, 0.0))), (D(), D())))
ps = ins.psolver_direct(setup)
u = ins.velocityfield(setup, lambda a, x, y, z: 0 * (x + y + z), 0.0, psolver=ps, doproject=False)
m = ins.RKMethods.RK44(); cache = ins.ode_method_cache(m, setup, ps)
st = ins.create_stepper(m, setup=setup, psolver=ps, u=u, t=0.0)
h = 0.9 * ins.get_cfl_timestep_(None, st.u, setup)
for _ in range(2): st = ins.timestep_(m, st, h, cache=cache)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): st = ins.timestep_(m, st, h, cache=cache)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"walls {n}^3: {dt*1e3:.2f} ms/step  {n**3/dt/1e6:.0f} M cells/s")
