#!/usr/bin/env python3
"""Body-force-driven channel flow: periodic in the streamwise and spanwise directions, tanh-stretched no-slip walls (the setting of
examples/TurbulentChannel.jl, which has its walls in z; here they are in y).  In either orientation the direct Poisson solver runs the two
periodic uniform directions in Fourier modes and only the wall-normal direction through its dense eigenvectors (csrc/ins_fdm.hip).
    python examples/TurbulentChannel.py n=64 tend=1"""
import numpy as np

import _common  # noqa: F401
import ins_amd as ins


def main(n=32, tend=0.05, Re=6000.0, seed=0, verbose=True):
    x = (np.linspace(0.0, 4.0, 4 * n + 1), ins.tanh_grid(0.0, 1.0, n), np.linspace(0.0, 1.0, n + 1))
    per, wall = (ins.PeriodicBC(), ins.PeriodicBC()), (ins.DirichletBC(), ins.DirichletBC())

    def bodyforce(a, x, y, z, t):  # drives the Poiseuille profile, with a weak spanwise stirring
        return (a == 0) * 10 * 4 * y * (1 - y) + (a == 2) * np.sin(10 * np.pi * x) / 5 + 0 * (x + y + z)

    setup = ins.Setup(x=x, boundary_conditions=(per, wall, per), Re=Re, bodyforce=bodyforce, issteadybodyforce=True)
    psolver = ins.default_psolver(setup)
    rng = np.random.default_rng(seed)

    def u0(a, x, y, z):
        if a == 0:
            return 4 * y * (1 - y) + 0 * (x + z)
        if a == 2:
            return np.sin(10 * np.pi * x) * np.sin(5 * np.pi * y) / 10 + 0 * z
        return 0.05 * rng.standard_normal(np.broadcast_shapes(x.shape, y.shape, z.shape))

    ustart = ins.velocityfield(setup, u0, psolver=psolver)
    procs = dict(log=ins.timelogger(nupdate=10)) if verbose else {}
    (u, _, t), _ = ins.solve_unsteady(setup=setup, tlims=(0.0, tend), ustart=ustart, psolver=psolver, cfl=0.5, processors=procs)
    up = ins.to_numpy(ins.interpolate_u_p(u, setup))
    sl = tuple(slice(lo, hi) for lo, hi in setup.grid.Ip)
    prof = up[sl][..., 0].mean(axis=(0, 2))  # mean streamwise velocity over x and z
    return dict(t=t, profile=prof, bulk=float(prof.mean()), maxdiv=ins.max_abs_divergence(u, setup), psolver=type(psolver).__name__,
                E=ins.total_kinetic_energy(u, setup))


if __name__ == "__main__":
    r = main(**_common.cli(dict(n=32, tend=0.05, Re=6000.0, seed=0)))
    print(f"{r['psolver']}: t = {r['t']:.3f}, bulk velocity {r['bulk']:.4f}, centreline {r['profile'][len(r['profile']) // 2]:.4f}, max|div u| = {r['maxdiv']:.2e}")
