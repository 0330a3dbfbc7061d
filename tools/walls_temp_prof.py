#!/usr/bin/env python3
"""Profile target: all-walls uniform box with the temperature equation, RK44 steps: tools/walls_temp_prof.py n steps"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
n = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
x = (np.linspace(0, 1, n + 1), np.linspace(0, 1, n + 1), np.linspace(0, 2, n + 1))
D, S = ins.DirichletBC, ins.SymmetricBC
T = ins.temperature_equation(Pr=0.71, Ra=1e6, Ge=1.0, boundary_conditions=((S(), S()),) * 3, gdir=2)
setup = ins.Setup(x=x, boundary_conditions=((D(), D()),) * 3, temperature=T)
ps = ins.psolver_direct(setup)
u = ins.velocityfield(setup, lambda a, x, y, z: 0 * (x + y + z), 0.0, psolver=ps, doproject=False)
temp = ins.temperaturefield(setup, lambda x, y, z: (1 + np.sin(np.pi * x / 20) * np.sin(np.pi * y) > z) * 1.0)
m = ins.RKMethods.RK44(); cache = ins.ode_method_cache(m, setup, ps)
st = ins.create_stepper(m, setup=setup, psolver=ps, u=u, temp=temp, t=0.0)
for _ in range(steps): st = ins.timestep_(m, st, 1e-3, cache=cache)
torch.cuda.synchronize(); print("done")
