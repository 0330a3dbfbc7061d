#!/usr/bin/env python3
"""Closure force on wall-bounded / stretched 256³ boxes: one generalised kernel (csrc/ins_smagforce.hip, GEN) against the three kernels."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
from ins_amd import _lib
n = 256
D, P = ins.DirichletBC, ins.PeriodicBC
cases = {"walls uniform": ((np.linspace(0, 1, n + 1),) * 3, ((D(), D()),) * 3),
         "cavity (cosine x, y; periodic z)": ((ins.cosine_grid(0, 1, n), ins.cosine_grid(0, 1, n), np.linspace(0, 1, n + 1)), ((D(), D()), (D(), D()), (P(), P())))}
for name, (x, bc) in cases.items():
    setup = ins.Setup(x=x, Re=1000.0, boundary_conditions=bc)
    u = ins.apply_bc_u(ins.from_numpy(setup, np.asfortranarray(np.random.default_rng(0).standard_normal(setup.grid.N + (3,)))), 0.0, setup)
    m = ins.smagorinsky_closure(setup)
    def t(label, **opts):
        with _lib.options(**opts):
            for _ in range(3): m(u, 0.17)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): m(u, 0.17)
            e1.record(); torch.cuda.synchronize()
        print(f"{name}: {label}: {e0.elapsed_time(e1) / 20:.3f} ms", flush=True)
    t("three kernels", INS_DISABLE_SMAGFORCE_GEN=1)
    t("one kernel")
    t("one kernel zc=64", INS_SMAGFORCE_ZC=64)
    t("one kernel, barrier", INS_SMAGFORCE_BAR=1)
