#!/usr/bin/env python3
"""Flow through an actuator disk: an unsteady inflow on the left, open (pressure) boundaries elsewhere, a steady body force inside a thin
disk (the setting of examples/Actuator2D.jl), RK44P2, velocity-norm / pressure / vorticity observers at the end.
    python examples/Actuator2D.py n=40 tend=12"""
import numpy as np

import _common  # noqa: F401
import ins_amd as ins


def main(n=24, tend=2.0, dt=0.05, Re=100.0, verbose=True):
    x = (np.linspace(0.0, 10.0, 5 * n + 1), np.linspace(-2.0, 2.0, 2 * n + 1))

    def inflow(a, x, y, t):
        return np.sin(np.pi * (np.sin(np.pi * t / 6) / 6 + (a == 0) / 2)) + 0 * (x + y)

    bcs = ((ins.DirichletBC(inflow), ins.PressureBC()), (ins.PressureBC(), ins.PressureBC()))
    xc, yc, D, δ, C = 2.0, 0.0, 1.0, 0.11, 0.2  # disk centre, diameter, thickness, thrust coefficient
    c = C / (D * δ)

    def bodyforce(a, x, y, t):
        return -c * (a == 0) * ((np.abs(x - xc) <= δ / 2) & (np.abs(y - yc) <= D / 2))

    setup = ins.Setup(x=x, Re=Re, boundary_conditions=bcs, bodyforce=bodyforce, issteadybodyforce=True)
    psolver = ins.default_psolver(setup)
    ustart = ins.velocityfield(setup, lambda a, x, y: inflow(a, x, y, 0.0), psolver=psolver)
    procs = dict(log=ins.timelogger(nupdate=24)) if verbose else {}
    (u, _, t), _ = ins.solve_unsteady(setup=setup, tlims=(0.0, tend), ustart=ustart, method=ins.RKMethods.RK44P2(), Δt=dt, psolver=psolver,
                                     processors=procs)
    state = dict(u=u, temp=None, t=t, n=0)
    fields = {k: ins.observefield(state, setup=setup, fieldname=k, psolver=psolver).value for k in ("velocitynorm", "pressure", "vorticity")}
    j0 = fields["velocitynorm"].shape[1] // 2
    xs = np.asarray(setup.grid.xp[0][setup.grid.Ip[0][0] : setup.grid.Ip[0][1]])
    wake = float(fields["velocitynorm"][np.searchsorted(xs, 4.0), j0])
    free = float(fields["velocitynorm"][np.searchsorted(xs, 4.0), 2])
    return dict(wake=wake, free=free, maxdiv=ins.max_abs_divergence(u, setup), fields=fields, psolver=type(psolver).__name__)


if __name__ == "__main__":
    r = main(**_common.cli(dict(n=24, tend=2.0, dt=0.05, Re=100.0)))
    print(f"{r['psolver']}: |u| two diameters behind the disk = {r['wake']:.3f} (free stream {r['free']:.3f}); max|div u| = {r['maxdiv']:.2e}")
