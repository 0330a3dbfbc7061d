#!/usr/bin/env python3
"""A short chained RK44 run for rocprofv3: tools/prof_step.py N [steps]  (TGV3D N^3; 1 call of `timesteps_` with `steps` steps after 2 warm-up steps,
then 3 plain momentum launches: K1 alone)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ins_amd as ins
from step_lab import tgv3d

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
setup = ins.Setup(x=(np.linspace(0.0, 1.0, n + 1),) * 3, Re=1000.0)
ps = ins.psolver_spectral(setup)
u0 = ins.velocityfield(setup, tgv3d, 0.0, psolver=ps)
m = ins.RKMethods.RK44()
cache = ins.ode_method_cache(m, setup, ps)
dt = 1e-3 if n <= 256 else 2.5e-4
st = ins.create_stepper(m, setup=setup, psolver=ps, u=u0, t=0.0)
st = ins.timesteps_(m, st, dt, 2, cache=cache)
st = ins.timesteps_(m, st, dt, steps, cache=cache)
F = ins.vectorfield(setup)
for _ in range(3):
    ins.momentum_(F, st.u, None, 0.0, setup)
torch.cuda.synchronize()
