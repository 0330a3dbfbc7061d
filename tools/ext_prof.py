#!/usr/bin/env python3
"""Profile target: a few steps of the extended native stage loop: tools/ext_prof.py n temp|smag|both [steps]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
n, mode = int(sys.argv[1]), sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
x = (np.linspace(0, 1, n + 1),) * 3
per = (ins.PeriodicBC(), ins.PeriodicBC())
T = ins.temperature_equation(Pr=0.71, Ra=1e6, Ge=0.1, boundary_conditions=(per, per, per)) if mode in ("temp", "both") else None
setup = ins.Setup(x=x, Re=1000.0, temperature=T)
if mode in ("smag", "both"):
    setup.closure_model = ins.smagorinsky_closure(setup)
ps = ins.psolver_spectral(setup)
u = ins.random_field(setup, kp=4, psolver=ps, seed=0)
temp = None if T is None else ins.temperaturefield(setup, lambda x, y, z: 0.5 + 0.1 * np.sin(2 * np.pi * x) + 0 * (y + z))
m = ins.RKMethods.RK44(); cache = ins.ode_method_cache(m, setup, ps)
st = ins.create_stepper(m, setup=setup, psolver=ps, u=u, temp=temp, t=0.0)
for _ in range(steps): st = ins.timestep_(m, st, 1e-4, θ=0.1, cache=cache)
torch.cuda.synchronize()
print("done", float(st.u.abs().max()))
