#!/usr/bin/env python3
"""Planar mixing layer: a perturbed tanh inflow profile on the left (time-dependent Dirichlet data), pressure boundaries elsewhere (the
setting of examples/PlanarMixing2D.jl), RK44P2 with adaptive time steps and the direct Poisson solver.
    python examples/PlanarMixing2D.py n=64 tend=100"""
import numpy as np

import _common  # noqa: F401
import ins_amd as ins


def main(n=32, tend=10.0, Re=500.0, verbose=True):
    dU, Ubar = 1.0, 1.0
    eps, nn, om = (0.082 * Ubar, 0.012 * Ubar), (0.4 * np.pi, 0.3 * np.pi), (0.22, 0.11)

    def U(a, x, y, t):
        if a != 0:
            return 0 * (x + y)
        pert = sum(e * (1 - np.tanh(y / 2) ** 2) * np.cos(k * y) * np.sin(w * t) for e, k, w in zip(eps, nn, om))
        return 1.0 + dU / 2 * np.tanh(2 * y) + pert + 0 * x

    bcs = ((ins.DirichletBC(U), ins.PressureBC()), (ins.PressureBC(), ins.PressureBC()))
    x = (np.linspace(0.0, 256.0, 4 * n), np.linspace(-32.0, 32.0, n))  # LinRange(a, b, m): m points
    setup = ins.Setup(x=x, Re=Re, boundary_conditions=bcs)
    psolver = ins.psolver_direct(setup)
    ustart = ins.velocityfield(setup, lambda a, x, y: U(a, x, y, 0.0), psolver=psolver)
    procs = dict(log=ins.timelogger(nupdate=100)) if verbose else {}
    (u, _, t), _ = ins.solve_unsteady(setup=setup, tlims=(0.0, tend), ustart=ustart, method=ins.RKMethods.RK44P2(), psolver=psolver, processors=procs)
    up = ins.to_numpy(ins.interpolate_u_p(u, setup))
    sl = tuple(slice(lo, hi) for lo, hi in setup.grid.Ip)
    return dict(t=t, ulo=float(up[sl][:, 1, 0].mean()), uhi=float(up[sl][:, -2, 0].mean()), vmax=float(np.abs(up[sl][..., 1]).max()),
                maxdiv=ins.max_abs_divergence(u, setup))


if __name__ == "__main__":
    r = main(**_common.cli(dict(n=32, tend=10.0, Re=500.0)))
    print(f"t = {r['t']:.2f}: <u> slow side {r['ulo']:.3f}, fast side {r['uhi']:.3f}, max|v| = {r['vmax']:.4f}, max|div u| = {r['maxdiv']:.2e}")
