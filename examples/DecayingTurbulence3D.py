#!/usr/bin/env python3
"""Decaying homogeneous isotropic turbulence from a random solenoidal field (the setting of examples/DecayingTurbulence3D.jl; BASELINE
config 3 in fp64): adaptive CFL time stepping, spectrum before and after, optional VTK snapshots.
    python examples/DecayingTurbulence3D.py n=128 tend=0.1 vtk=out"""
import numpy as np

import _common  # noqa: F401
import ins_amd as ins


def main(n=64, tend=0.05, Re=4000.0, kp=10, cfl=0.9, seed=0, vtk="", verbose=True):
    setup = ins.Setup(x=(np.linspace(0.0, 1.0, n + 1),) * 3, Re=Re)
    psolver = ins.psolver_spectral(setup)
    ustart = ins.random_field(setup, 0.0, kp=kp, psolver=psolver, seed=seed)
    spec0 = ins.observespectrum(dict(u=ustart, temp=None, t=0.0, n=0), setup=setup)
    procs = {}
    if verbose:
        procs["log"] = ins.timelogger(nupdate=5)
    if vtk:
        procs["vtk"] = ins.vtk_writer(setup=setup, nupdate=10, dir=vtk, fieldnames=("velocity", "Qfield"))
    (u, _, t), out = ins.solve_unsteady(setup=setup, tlims=(0.0, tend), ustart=ustart, cfl=cfl, psolver=psolver, processors=procs)
    spec1 = ins.observespectrum(dict(u=u, temp=None, t=t, n=0), setup=setup)
    return dict(κ=spec0["κ"], ehat0=spec0["ehat"].value, ehat1=spec1["ehat"].value, E0=ins.total_kinetic_energy(ustart, setup),
                E1=ins.total_kinetic_energy(u, setup), maxdiv=ins.max_abs_divergence(u, setup), t=t)


if __name__ == "__main__":
    r = main(**_common.cli(dict(n=64, tend=0.05, Re=4000.0, kp=10, cfl=0.9, seed=0, vtk="")))
    print(f"E: {r['E0']:.5e} -> {r['E1']:.5e} at t = {r['t']:.4f}; spectrum peak at κ = {r['κ'][np.argmax(r['ehat0'])]}; max|div u| = {r['maxdiv']:.2e}")
