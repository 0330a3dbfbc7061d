#!/usr/bin/env python3
"""Taylor-Green vortex in a periodic unit box (the setting of the reference's examples/TaylorGreenVortex3D.jl; BASELINE config 2):
RK44 + spectral pressure projection, kinetic energy history, energy spectrum at the end.
    python examples/TaylorGreenVortex3D.py n=256 tend=0.1 dt=1e-3"""
import time

import numpy as np

import _common  # noqa: F401
import ins_amd as ins


def main(n=64, tend=0.05, dt=1e-3, Re=1000.0, nupdate=10, verbose=True):
    setup = ins.Setup(x=(np.linspace(0.0, 1.0, n + 1),) * 3, Re=Re)
    psolver = ins.psolver_spectral(setup)

    def u0(a, x, y, z):
        if a == 0:
            return np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y) * np.sin(2 * np.pi * z) / 2
        if a == 1:
            return -np.cos(2 * np.pi * x) * np.sin(2 * np.pi * y) * np.sin(2 * np.pi * z) / 2
        return 0 * (x + y + z)

    ustart = ins.velocityfield(setup, u0, psolver=psolver)
    energy = []

    def ehist(state):
        state.on(lambda s: s["n"] % nupdate == 0 and energy.append((s["t"], ins.total_kinetic_energy(s["u"], setup))))
        return energy

    procs = dict(ehist=ins.processor(ehist))
    if verbose:
        procs["log"] = ins.timelogger(nupdate=nupdate)
    t0 = time.time()
    (u, _, t), out = ins.solve_unsteady(setup=setup, tlims=(0.0, tend), ustart=ustart, Δt=dt, psolver=psolver, processors=procs)
    wall = time.time() - t0
    spec = ins.observespectrum(dict(u=u, temp=None, t=t, n=0), setup=setup)
    return dict(energy=energy, ehat=spec["ehat"].value, κ=spec["κ"], maxdiv=ins.max_abs_divergence(u, setup), wall=wall, u=u, setup=setup)


if __name__ == "__main__":
    r = main(**_common.cli(dict(n=64, tend=0.05, dt=1e-3, Re=1000.0, nupdate=10)))
    print(f"E(0+) = {r['energy'][0][1]:.6e}  E(end) = {r['energy'][-1][1]:.6e}  max|div u| = {r['maxdiv']:.2e}  wall = {r['wall']:.2f} s")
