#!/usr/bin/env python3
"""Rayleigh-Bénard convection between a hot bottom plate (T = 1) and a cold top plate (T = 0), insulated side walls (the setting of
examples/RayleighBenard2D.jl): temperature equation with viscous heating, buoyancy in y, tanh-stretched grid, Nusselt numbers at both plates.
    python examples/RayleighBenard2D.py n=100 tend=20 dt=1e-2"""
import numpy as np

import _common  # noqa: F401
import ins_amd as ins


def main(n=48, tend=2.0, dt=1e-2, Pr=0.71, Ra=1e6, Ge=1.0, verbose=True):
    temperature = ins.temperature_equation(
        Pr=Pr, Ra=Ra, Ge=Ge, dodissipation=True, gdir=1, nondim_type=1,
        boundary_conditions=((ins.SymmetricBC(), ins.SymmetricBC()), (ins.DirichletBC(1.0), ins.DirichletBC(0.0))))
    x = (ins.tanh_grid(0.0, 2.0, 2 * n, 1.2), ins.tanh_grid(0.0, 1.0, n, 1.2))
    walls = (ins.DirichletBC(), ins.DirichletBC())
    setup = ins.Setup(x=x, boundary_conditions=(walls, walls), temperature=temperature)  # Re = 1/α1
    psolver = ins.default_psolver(setup)
    ustart = ins.velocityfield(setup, lambda a, x, y: 0 * (x + y), psolver=psolver)
    tempstart = ins.temperaturefield(setup, lambda x, y: 0.5 + np.maximum(np.sin(20 * np.pi * x) / 100, 0) + 0 * y)
    g = setup.grid
    dy1, dy2 = g.Δu[1][0], g.Δu[1][-2]
    dx = np.asarray(g.Δ[0])
    nusselt = []

    def watch(state):
        def nu(s):
            if s["n"] % 10:
                return
            T = ins.to_numpy(s["temp"])
            lo = np.sum((-(T[:, 1] - T[:, 0]) / dy1 * dx)[1:-1])
            hi = np.sum((-(T[:, -2] - T[:, -3]) / dy2 * dx)[1:-1])
            nusselt.append((s["t"], lo, hi))

        state.on(nu)
        return nusselt

    procs = dict(nusselt=ins.processor(watch))
    if verbose:
        procs["log"] = ins.timelogger(nupdate=50)
    (u, temp, t), out = ins.solve_unsteady(setup=setup, tlims=(0.0, tend), ustart=ustart, tempstart=tempstart, Δt=dt, psolver=psolver, processors=procs)
    T = ins.to_numpy(temp)
    return dict(nusselt=nusselt, Tmean=float(T[1:-1, 1:-1].mean()), Tmin=float(T[1:-1, 1:-1].min()), Tmax=float(T[1:-1, 1:-1].max()),
                E=ins.total_kinetic_energy(u, setup), maxdiv=ins.max_abs_divergence(u, setup), Re=setup.Re)


if __name__ == "__main__":
    r = main(**_common.cli(dict(n=48, tend=2.0, dt=1e-2, Pr=0.71, Ra=1e6, Ge=1.0)))
    t, lo, hi = r["nusselt"][-1]
    print(f"Re = {r['Re']:.1f}; t = {t:.2f}: Nu(bottom) = {lo:.3f}, Nu(top) = {hi:.3f}; T in [{r['Tmin']:.3f}, {r['Tmax']:.3f}]; E = {r['E']:.3e}")
