#!/usr/bin/env python3
"""Config 5 (LidDrivenCavity3D-shaped: cosine-stretched Dirichlet x/y, periodic z, psolver_direct) step loop for rocprofv3."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
x = (ins.cosine_grid(0.0, 1.0, n), ins.cosine_grid(0.0, 1.0, n), np.linspace(-0.2, 0.2, n + 1))
D, P = ins.DirichletBC, ins.PeriodicBC
setup = ins.Setup(x=x, Re=1000.0, boundary_conditions=((D(), D()), (D(), D((1.0, 0.2, 0.0))), (P(), P())))
ps = ins.psolver_direct(setup)
u = ins.velocityfield(setup, lambda a, x, y, z: 0 * (x + y + z), 0.0, psolver=ps, doproject=False)
m = ins.RKMethods.RK44(); cache = ins.ode_method_cache(m, setup, ps)
st = ins.create_stepper(m, setup=setup, psolver=ps, u=u, t=0.0)
h = 0.9 * ins.get_cfl_timestep_(None, st.u, setup)
for _ in range(2): st = ins.timestep_(m, st, h, cache=cache)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): st = ins.timestep_(m, st, h, cache=cache)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"cavity {n}^3: {dt*1e3:.2f} ms/step  {n**3/dt/1e6:.0f} M cells/s")
