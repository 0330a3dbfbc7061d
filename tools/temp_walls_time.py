#!/usr/bin/env python3
"""ms/step of a wall-bounded box with the temperature equation (examples/RayleighTaylor3D.jl shape: Dirichlet walls, Symmetric temperature
sides, uniform grid, direct solver) through the native extended stage loop vs the host-driven loop: tools/temp_walls_time.py n"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
n = int(sys.argv[1])
x = (np.linspace(0, 1, n + 1), np.linspace(0, 1, n + 1), np.linspace(0, 2, n + 1))
D, S = ins.DirichletBC, ins.SymmetricBC
T = ins.temperature_equation(Pr=0.71, Ra=1e6, Ge=1.0, boundary_conditions=((S(), S()),) * 3, gdir=2)
for temperature in (None, T):
    setup = ins.Setup(x=x, Re=1000.0 if temperature is None else None, boundary_conditions=((D(), D()),) * 3, temperature=temperature)
    ps = ins.psolver_direct(setup)
    u = ins.velocityfield(setup, lambda a, x, y, z: 0 * (x + y + z), 0.0, psolver=ps, doproject=False)
    temp = None if temperature is None else ins.temperaturefield(setup, lambda x, y, z: (1 + np.sin(np.pi * x / 20) * np.sin(np.pi * y) > z) * 1.0)
    m = ins.RKMethods.RK44(); cache = ins.ode_method_cache(m, setup, ps)
    st = ins.create_stepper(m, setup=setup, psolver=ps, u=u, temp=temp, t=0.0)
    for _ in range(2): st = ins.timestep_(m, st, 1e-3, cache=cache)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): st = ins.timestep_(m, st, 1e-3, cache=cache)
    torch.cuda.synchronize()
    print(f"walls n={n} temperature={'yes' if temperature else 'no'} host_loop={bool(os.environ.get('INS_HOST_STAGE_LOOP'))}: {(time.perf_counter()-t0)*200:.3f} ms/step", flush=True)
