#!/usr/bin/env python3
"""Sanity + timing of the BASELINE configs that are not the bench default:
   config 3: DecayingTurbulence3D 512^3 (random field, spectral)   config 5: LidDrivenCavity3D stretched + Dirichlet + CG."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins

def run_cavity(n, solver="direct"):
    x = (ins.cosine_grid(0.0, 1.0, n), ins.cosine_grid(0.0, 1.0, n), np.linspace(-0.2, 0.2, n + 1))
    D, P = ins.DirichletBC, ins.PeriodicBC
    setup = ins.Setup(x=x, Re=1000.0, boundary_conditions=((D(), D()), (D(), D((1.0, 0.2, 0.0))), (P(), P())))
    t0 = time.perf_counter()
    ps = ins.psolver_direct(setup) if solver == "direct" else ins.psolver_cg(setup, bordered=True)  # default of the example / reltol sqrt(eps) as pressure.jl:212
    t_setup = time.perf_counter() - t0
    u = ins.velocityfield(setup, lambda a, x, y, z: 0 * (x + y + z), 0.0, psolver=ps, doproject=False)
    m = ins.RKMethods.RK44(); cache = ins.ode_method_cache(m, setup, ps)
    st = ins.create_stepper(m, setup=setup, psolver=ps, u=u, t=0.0)
    h = 0.9 * ins.get_cfl_timestep_(None, st.u, setup)   # adaptive step of solve_unsteady (solver.jl:57-61): u = 0 -> diffusive limit
    st = ins.timestep_(m, st, h, cache=cache); torch.cuda.synchronize()
    t0 = time.perf_counter(); its = []
    for _ in range(3):
        h = 0.9 * ins.get_cfl_timestep_(None, st.u, setup)
        st = ins.timestep_(m, st, h, cache=cache); its.append(ps.last_info()[0])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"   dt_cfl = {h:.3e}", end=" ")
    print(f"config5 cavity {n}^3 psolver_{solver} (setup {t_setup:.2f} s): {dt*1e3:.1f} ms/step, {n**3/dt/1e6:.1f} M cells/s, CG its (last solve of step) {its}, "
          f"max|div| {ins.max_abs_divergence(st.u, setup):.2e}, finite {bool(torch.isfinite(st.u).all())}", flush=True)

def run_turb(n):
    setup = ins.Setup(x=(np.linspace(0, 1, n + 1),) * 3, Re=4000.0)
    ps = ins.psolver_spectral(setup)
    u = ins.random_field(setup, kp=10, A=1.0, seed=0, psolver=ps)
    e0 = ins.total_kinetic_energy(u, setup)
    m = ins.RKMethods.RK44(); cache = ins.ode_method_cache(m, setup, ps)
    st = ins.create_stepper(m, setup=setup, psolver=ps, u=u, t=0.0)
    st = ins.timestep_(m, st, 2.5e-4, cache=cache); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): st = ins.timestep_(m, st, 2.5e-4, cache=cache)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"config3 decaying turbulence {n}^3: {dt*1e3:.1f} ms/step, {n**3/dt/1e6:.0f} M cells/s, E0 {e0:.4f} -> {ins.total_kinetic_energy(st.u, setup):.4f}, "
          f"max|div|*dx {ins.max_abs_divergence(st.u, setup)/n:.2e}", flush=True)

if __name__ == "__main__":
    for n in (64, 128, 256): run_cavity(n, "direct")
    for n in (64, 128): run_cavity(n, "cg")
    run_turb(512)
