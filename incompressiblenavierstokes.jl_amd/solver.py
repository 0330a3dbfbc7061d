"""`solve_unsteady` and the CFL time step (solver.jl)."""
import ctypes as C
import math
import os

from . import _lib
from .pressure import default_psolver
from .setup import copyfield
from .time_steppers import RKMethods, create_stepper, ode_method_cache, timestep_, timesteps_


def get_state(stepper):
    """solver.jl:95-98"""
    return dict(u=stepper.u, temp=stepper.temp, t=stepper.t, n=stepper.n)


def get_cfl_timestep_(buf, u, setup):
    """Get proposed maximum time step for convection and diffusion terms (solver.jl:101-125).
    `buf` is accepted for signature parity; the reduction scratch is internal.  Blocking."""
    out = C.c_double()
    _lib.call("ins_cfl_timestep_f64", setup.handle, setup.Re, setup.ptr(u, True), C.byref(out), setup.stream)
    return out.value


def solve_unsteady(*, setup, tlims, ustart, tempstart=None, method=None, psolver=None, Δt=None, Δt_min=None,
                   cfl=0.9, n_adapt_Δt=1, docopy=True, processors=None, θ=None, cache=None):
    """Solve unsteady problem using `method` (solver.jl:18-92).

    `processors` is a dict name -> object with `initialize(state_getter)` / `finalize(init, state_getter)`
    and an optional `on_step(state)`; returns `((u, temp, t), outputs)` like the reference."""
    method = method or RKMethods.RK44()
    psolver = psolver or default_psolver(setup)
    cache = cache or ode_method_cache(method, setup, psolver)
    processors = processors or {}
    if docopy:
        ustart = copyfield(ustart)
        tempstart = None if tempstart is None else copyfield(tempstart)
    tstart, tend = tlims
    isadaptive = Δt is None
    stepper = create_stepper(method, setup=setup, psolver=psolver, u=ustart, temp=tempstart, t=tstart)
    state = {"value": get_state(stepper)}
    initialized = {k: v.initialize(lambda: state["value"]) for k, v in processors.items()}

    def fire():
        state["value"] = get_state(stepper)
        for v in processors.values():
            if hasattr(v, "on_step"):
                v.on_step(state["value"])

    if isadaptive:
        while stepper.t < tend:
            if stepper.n % n_adapt_Δt == 0:
                Δt = cfl * get_cfl_timestep_(None, stepper.u, setup)
                Δt = Δt if Δt_min is None else max(Δt, Δt_min)
            Δt = min(Δt, tend - stepper.t)
            stepper = timestep_(method, stepper, Δt, θ=θ, cache=cache)
            fire()
    else:
        nstep = int(round((tend - tstart) / Δt))
        Δt = (tend - tstart) / nstep
        if not processors:  # nobody looks at the intermediate states: the whole loop is one native call
            stepper = timesteps_(method, stepper, Δt, nstep, θ=θ, cache=cache)
            fire()
        else:
            # processors that act every `nupdate` steps only (timelogger, fieldsaver, vtk_writer say so; a user-made processor has nupdate = 1) never look at
            # the states in between: those steps run as one native call (chained steps), and the state fires at the multiples of gcd(nupdate...) — every
            # step a processor acts on is among them.  INS_NO_PROCESSOR_BATCH=1: one call and one state update per step, as the reference's loop.
            g = 0
            for v in processors.values():
                g = math.gcd(g, int(getattr(v, "nupdate", 1)))
            if g <= 1 or os.environ.get("INS_NO_PROCESSOR_BATCH"):
                g = 1
            done = 0
            while done < nstep:
                k = min(g - (stepper.n % g), nstep - done)
                stepper = timesteps_(method, stepper, Δt, k, θ=θ, cache=cache) if k > 1 else timestep_(method, stepper, Δt, θ=θ, cache=cache)
                done += k
                fire()
    outputs = {k: processors[k].finalize(initialized[k], lambda: state["value"]) for k in processors}
    return (stepper.u, stepper.temp, stepper.t), outputs
