#!/usr/bin/env python3
"""Lid-driven cavity: cosine-stretched walls in x and y, periodic z, moving lid (the setting of examples/LidDrivenCavity3D.jl; BASELINE
config 5) with the direct (fast-diagonalisation) Poisson solver.
    python examples/LidDrivenCavity3D.py n=64 tend=0.2"""
import numpy as np

import _common  # noqa: F401
import ins_amd as ins


def main(n=32, nz=0, tend=0.1, dt=0.0, Re=1000.0, cfl=0.2, verbose=True):
    """dt = 0: adaptive steps from the CFL / diffusion limit (the wall cells of a cosine grid shrink like n⁻², so a fixed Δt that is
    stable at n = 25 is not at n = 64).  cfl = 0.2: the reference's diffusive bound Re Δmin²/2 looks at one direction at a time, and in the
    corner cells two directions are at the bound together and the half-width wall cells double the stiffness (RK44 needs Δt ν Σ 4/Δ² < 2.78; measured: 0.4 blows up at n = 64, 0.2 and 0.1 agree)."""
    nz = nz or max(n // 4, 4)
    x = (ins.cosine_grid(0.0, 1.0, n), ins.cosine_grid(0.0, 1.0, n), np.linspace(-0.2, 0.2, nz + 1))
    lid = ins.DirichletBC((1.0, 0.0, 0.2))
    bcs = ((ins.DirichletBC(), ins.DirichletBC()), (ins.DirichletBC(), lid), (ins.PeriodicBC(), ins.PeriodicBC()))
    setup = ins.Setup(x=x, boundary_conditions=bcs, Re=Re)
    psolver = ins.default_psolver(setup)  # psolver_direct on this grid
    ustart = ins.velocityfield(setup, lambda a, x, y, z: 0 * (x + y + z), psolver=psolver)
    procs = dict(log=ins.timelogger(nupdate=20)) if verbose else {}
    (u, _, t), _ = ins.solve_unsteady(setup=setup, tlims=(0.0, tend), ustart=ustart, Δt=dt or None, cfl=cfl, psolver=psolver, processors=procs)
    up = ins.to_numpy(ins.interpolate_u_p(u, setup))
    return dict(E=ins.total_kinetic_energy(u, setup), maxdiv=ins.max_abs_divergence(u, setup), umax=float(np.abs(up).max()), t=t, psolver=type(psolver).__name__)


if __name__ == "__main__":
    r = main(**_common.cli(dict(n=32, nz=0, tend=0.1, dt=0.0, Re=1000.0, cfl=0.2)))
    print(f"{r['psolver']}: E = {r['E']:.5e}, max|u| at pressure points = {r['umax']:.3f}, max|div u| = {r['maxdiv']:.2e}")
