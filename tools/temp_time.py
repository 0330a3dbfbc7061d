#!/usr/bin/env python3
"""ms/step with the temperature equation (host-driven stage loop) vs the plain native step: tools/temp_time.py n"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
n = int(sys.argv[1])
x = (np.linspace(0, 1, n + 1),) * 3
per = (ins.PeriodicBC(), ins.PeriodicBC())
T = ins.temperature_equation(Pr=0.71, Ra=1e6, Ge=0.1, boundary_conditions=(per, per, per))
for temperature, closure in ((None, False), (T, False), (None, True)):
    setup = ins.Setup(x=x, Re=1000.0, temperature=temperature)
    if closure:
        setup.closure_model = ins.smagorinsky_closure(setup)
    ps = ins.psolver_spectral(setup)
    u = ins.random_field(setup, kp=4, psolver=ps, seed=0)
    temp = None if temperature is None else ins.temperaturefield(setup, lambda x, y, z: 0.5 + 0.1 * np.sin(2 * np.pi * x) + 0 * (y + z))
    m = ins.RKMethods.RK44(); cache = ins.ode_method_cache(m, setup, ps)
    st = ins.create_stepper(m, setup=setup, psolver=ps, u=u, temp=temp, t=0.0)
    for _ in range(2): st = ins.timestep_(m, st, 1e-4, θ=0.1, cache=cache)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): st = ins.timestep_(m, st, 1e-4, θ=0.1, cache=cache)
    torch.cuda.synchronize()
    print(f"n={n} temperature={'yes' if temperature else 'no'} closure={'smagorinsky' if closure else 'no'}: {(time.perf_counter()-t0)*100:.3f} ms/step", flush=True)
