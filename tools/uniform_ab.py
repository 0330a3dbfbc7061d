#!/usr/bin/env python3
"""Constant-record kernels vs table-driven kernels on a box that is uniform only up to rounding ([0, 2π]³): momentum and 3 RK44 steps.
    python tools/uniform_ab.py n save|compare   (run `save` with INS_UNIFORM_BITWISE=1, `compare` without)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
n, mode = int(sys.argv[1]), sys.argv[2]
setup = ins.Setup(x=(np.linspace(0, 2 * np.pi, n + 1),) * 3, Re=1000.0)
ps = ins.psolver_spectral(setup)
g = torch.Generator(device=setup.device).manual_seed(1)
u = ins.vectorfield(setup); u.copy_(torch.randn(u.shape, dtype=torch.float64, device=u.device, generator=g))
u = ins.project(u, setup, ps); ins.apply_bc_u_(u, 0.0, setup)
F = ins.momentum(u, None, 0.0, setup)
(v, _, _), _ = ins.solve_unsteady(setup=setup, tlims=(0.0, 3e-3), ustart=u, psolver=ps, Δt=1e-3)
torch.cuda.synchronize()
if mode == "save":
    torch.save({"F": F.cpu(), "v": v.cpu()}, "/tmp/uniform_ab.pt")
    print("saved; uniform_exact =", ins._lib.load().ins_grid_is_uniform_exact(setup.handle))
else:
    ref = torch.load("/tmp/uniform_ab.pt")
    dF = float((F.cpu() - ref["F"]).abs().max() / ref["F"].abs().max())
    dv = float(((v.cpu() - ref["v"]) ** 2).sum().sqrt() / (ref["v"] ** 2).sum().sqrt())
    print(f"n={n}: uniform_exact = {ins._lib.load().ins_grid_is_uniform_exact(setup.handle)}; momentum relmax diff {dF:.2e}; 3 RK44 steps rel L2 diff {dv:.2e}")
