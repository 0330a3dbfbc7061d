#!/usr/bin/env python3
"""Micro-benchmark of the momentum-RHS kernel (K1): generic vs flux-form variants, with parity checks.
Usage: python tools/k1_bench.py [n ...]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins

lib = ins._lib.load()
lib.ins_tune_flux3d.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]


def time_momentum(setup, u, F, iters=20):
    for _ in range(3):
        ins.momentum_(F, u, None, 0.0, setup)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ins.momentum_(F, u, None, 0.0, setup)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def rand_u(setup, seed=0):
    torch.manual_seed(seed)
    u = ins.vectorfield(setup)
    u.copy_(torch.randn(u.shape, dtype=torch.float64, device=u.device))
    ins.apply_bc_u_(u, 0.0, setup)
    return u


def parity():
    """flux-form variants vs the generic kernel on uniform, stretched-periodic and Dirichlet/mixed grids."""
    D, P, S, Pr = ins.DirichletBC, ins.PeriodicBC, ins.SymmetricBC, ins.PressureBC
    cases = {
        "uniform periodic 70x19x23": dict(x=(np.linspace(0, 1, 71), np.linspace(0, 2, 20), np.linspace(0, 1, 24))),
        "stretched periodic 130x9x12": dict(x=(ins.tanh_grid(0.0, 1.0, 130, 1.2), ins.cosine_grid(0.0, 1.0, 9), ins.tanh_grid(0.0, 2.0, 12))),
        "dirichlet stretched 66x21x17": dict(x=(ins.tanh_grid(0.0, 1.0, 66, 1.2), ins.tanh_grid(0.0, 1.0, 21, 1.1), ins.cosine_grid(0.0, 1.0, 17)),
                                             boundary_conditions=((D(), D()), (D(), D((1.0, 0.0, 0.2))), (D(), D()))),
        "mixed 63x10x9": dict(x=(ins.tanh_grid(0.0, 5.0, 63), ins.cosine_grid(0.0, 1.0, 10), ins.tanh_grid(0.0, 0.8, 9)),
                              boundary_conditions=((P(), P()), (D(), Pr()), (S(), S()))),
    }
    worst = 0.0
    for name, kw in cases.items():
        setup = ins.Setup(Re=100.0, **kw)
        u = rand_u(setup, 1)
        F = ins.vectorfield(setup)
        os.environ["INS_DISABLE_FAST3D"] = "1"
        ins.momentum_(F, u, None, 0.0, setup)
        ref = F.clone()
        del os.environ["INS_DISABLE_FAST3D"]
        for rows, zc, xw, km in ((1, 5, 4, 1), (2, 64, 4, 0), (3, 5, 2, 1), (4, 64, 1, 1), (2, 7, 1, 0), (4, 3, 4, 1)):
            if True:
                lib.ins_tune_flux3d(rows, zc, xw, 0, km)
                F.copy_(torch.full_like(F, 7.0))
                ins.momentum_(F, u, None, 0.0, setup)
                err = float((F - ref).abs().max() / ref.abs().max())
                worst = max(worst, err)
                flag = "" if err < 1e-12 else "   <-- FAIL"
                print(f"parity {name:32s} R={rows} zc={zc:3d} xw={xw} km={km}: relerr={err:.2e}{flag}", flush=True)
    print(f"parity worst relerr = {worst:.2e}", flush=True)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [256]
    parity()
    for n in sizes:
        setup = ins.Setup(x=(np.linspace(0, 1, n + 1),) * 3, Re=1000.0)
        u = rand_u(setup)
        F = ins.vectorfield(setup)
        os.environ["INS_DISABLE_FAST3D"] = "1"
        t = time_momentum(setup, u, F)
        ref = F.clone()
        gb = 48.0 * n**3 / 1e9
        print(f"n={n} generic          : {t:.4f} ms  {gb / t * 1e3:8.1f} GB/s", flush=True)
        del os.environ["INS_DISABLE_FAST3D"]
        for xw, nt in ((4, 0), (4, 1), (2, 0)):
            for rows in (1, 2, 3, 4):
                for zc in (2, 4, 8, 16):
                    lib.ins_tune_flux3d(rows, zc, xw, nt, 1)
                    F.zero_()
                    t = time_momentum(setup, u, F)
                    err = float((F - ref).abs().max() / ref.abs().max())
                    print(f"n={n} flux xw={xw} nt={nt} R={rows} zc={zc:3d}: {t:.4f} ms  {gb / t * 1e3:8.1f} GB/s  relerr={err:.2e}", flush=True)


if __name__ == "__main__":
    main()
