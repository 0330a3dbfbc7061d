#!/usr/bin/env python3
"""RK44 step time of the Float32 family on wall-bounded boxes (csrc/ins_f32g.hip) beside the fp64 step of the same setup:
tools/f32_walls_time.py [n] [steps]   cavity = LidDrivenCavity3D shape (cosine x cosine x periodic), walls = uniform all-Dirichlet box."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
from ins_amd import f32

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
D, P = ins.DirichletBC, ins.PeriodicBC
cases = {
    "cavity": ((ins.cosine_grid(0.0, 1.0, n), ins.cosine_grid(0.0, 1.0, n), np.linspace(-0.2, 0.2, n + 1)), ((D(), D()), (D(), D((1.0, 0.2, 0.0))), (P(), P()))),
    "walls": (tuple(np.linspace(0.0, 1.0, n + 1) for _ in range(3)), ((D(), D()), (D(), D((1.0, 0.0, 0.0))), (D(), D()))),
}
for name, (x, bc) in cases.items():
    setup = ins.Setup(x=x, Re=1000.0, boundary_conditions=bc)
    ps = ins.psolver_direct(setup)
    u = ins.velocityfield(setup, lambda a, x, y, z: 0 * (x + y + z), 0.0, psolver=ps, doproject=False)
    m = ins.RKMethods.RK44(); cache = ins.ode_method_cache(m, setup, ps)
    st = ins.create_stepper(m, setup=setup, psolver=ps, u=u, t=0.0)
    h = 0.05 * ins.get_cfl_timestep_(None, st.u, setup)  # the lid accelerates the fluid from rest: a step that stays stable over the timed steps
    for _ in range(2): st = ins.timestep_(m, st, h, cache=cache)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): st = ins.timestep_(m, st, h, cache=cache)
    torch.cuda.synchronize(); t64 = (time.perf_counter() - t0) / steps
    ps32 = f32.psolver_wrap32(setup, ps)
    c32 = f32.ERKCache32(m, setup, ps32)
    u32 = f32.to_f32(setup, u)
    f32.timesteps32_(c32, u32, h, 2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f32.timesteps32_(c32, u32, h, steps)
    torch.cuda.synchronize(); t32 = (time.perf_counter() - t0) / steps
    d = float((u32.double() - st.u).abs().max() / st.u.abs().max())
    print(f"{name} {n}^3: fp64 {t64*1e3:.2f} ms/step   fp32 {t32*1e3:.2f} ms/step   max|u32 - u64| / max|u64| = {d:.1e}", flush=True)
    del cache, c32, ps32, ps, st, u, u32
