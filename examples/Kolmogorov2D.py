#!/usr/bin/env python3
"""Kolmogorov flow: a periodic 2-D box driven by a steady sinusoidal body force (the setting of examples/Kolmogorov2D.jl), started from a
small random field; energy history and final spectrum.
    python examples/Kolmogorov2D.py n=256 tend=2"""
import numpy as np

import _common  # noqa: F401
import ins_amd as ins


def main(n=128, tend=0.5, dt=1e-3, Re=2000.0, verbose=True):
    axis = np.linspace(0.0, 1.0, n + 1)
    setup = ins.Setup(x=(axis, axis), Re=Re, bodyforce=lambda a, x, y, t: (a == 0) * 5 * np.sin(8 * np.pi * y) + 0 * x, issteadybodyforce=True)
    psolver = ins.psolver_spectral(setup)
    ustart = ins.random_field(setup, 0.0, A=1e-2, psolver=psolver)
    energy = []

    def ehist(state):
        state.on(lambda s: s["n"] % 10 == 0 and energy.append((s["t"], ins.total_kinetic_energy(s["u"], setup))))
        return energy

    procs = dict(ehist=ins.processor(ehist))
    if verbose:
        procs["log"] = ins.timelogger(nupdate=100)
    (u, _, t), out = ins.solve_unsteady(setup=setup, tlims=(0.0, tend), ustart=ustart, Δt=dt, psolver=psolver, processors=procs)
    spec = ins.observespectrum(dict(u=u, temp=None, t=t, n=0), setup=setup)
    up = ins.to_numpy(ins.interpolate_u_p(u, setup))
    # the forcing drives u(y) ~ sin(8πy): its Fourier amplitude in the mean x-velocity profile
    prof = up[1:-1, 1:-1, 0].mean(axis=0)
    y = np.asarray(setup.grid.xp[1][1:-1])
    amp = 2 * np.mean(prof * np.sin(8 * np.pi * y))
    return dict(energy=energy, ehat=spec["ehat"].value, κ=spec["κ"], forced_mode=float(amp), maxdiv=ins.max_abs_divergence(u, setup))


if __name__ == "__main__":
    r = main(**_common.cli(dict(n=128, tend=0.5, dt=1e-3, Re=2000.0)))
    print(f"E: {r['energy'][0][1]:.3e} -> {r['energy'][-1][1]:.3e}; amplitude of the forced mode in <u>(y): {r['forced_mode']:.4f}; max|div u| = {r['maxdiv']:.2e}")
