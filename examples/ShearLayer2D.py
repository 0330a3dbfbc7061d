#!/usr/bin/env python3
"""Double shear layer roll-up in a periodic 2π box (the setting of examples/ShearLayer2D.jl): vorticity extrema and enstrophy history.
    python examples/ShearLayer2D.py n=128 tend=8"""
import numpy as np

import _common  # noqa: F401
import ins_amd as ins


def main(n=64, tend=1.0, dt=0.01, Re=2000.0, verbose=True):
    x = (np.linspace(0.0, 2 * np.pi, n + 1),) * 2
    setup = ins.Setup(x=x, Re=Re)
    psolver = ins.psolver_spectral(setup)
    d, e = np.pi / 15, 0.05

    def u0(a, x, y):
        if a == 0:
            return np.where(y <= np.pi, np.tanh((y - np.pi / 2) / d), np.tanh((3 * np.pi / 2 - y) / d)) + 0 * x
        return e * np.sin(x) + 0 * y

    ustart = ins.velocityfield(setup, u0, psolver=psolver)
    hist = []

    def watch(state):
        w = ins.observefield(state, setup=setup, fieldname="vorticity")
        state.on(lambda s: s["n"] % 10 == 0 and hist.append((s["t"], float(np.abs(w.value).max()), float((w.value**2).mean()))))
        return hist

    procs = dict(watch=ins.processor(watch))
    if verbose:
        procs["log"] = ins.timelogger(nupdate=100)
    (u, _, t), _ = ins.solve_unsteady(setup=setup, tlims=(0.0, tend), ustart=ustart, Δt=dt, psolver=psolver, processors=procs)
    return dict(hist=hist, E=ins.total_kinetic_energy(u, setup), maxdiv=ins.max_abs_divergence(u, setup))


if __name__ == "__main__":
    r = main(**_common.cli(dict(n=64, tend=1.0, dt=0.01, Re=2000.0)))
    t, wmax, ens = r["hist"][-1]
    print(f"t = {t:.2f}: max|ω| = {wmax:.3f}, enstrophy = {ens:.4f}, E = {r['E']:.4f}, max|div u| = {r['maxdiv']:.2e}")
