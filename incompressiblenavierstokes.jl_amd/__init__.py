"""MI355X-native hot path of IncompressibleNavierStokes.jl behind the reference's own API names.

Julia's `op!` is spelled `op_` here; indices are 0-based; `backend = ...` becomes `device = ...`.
Everything numerical runs in hand-written HIP kernels + rocFFT inside libinship.so (C ABI:
include/ins_hip.h).  There is no CPU fallback: importing this package without the built library, or
creating a `Setup` without a HIP device, raises.
"""
from . import _lib
from . import f32  # noqa: F401  (the `_f32` entry-point family, T = Float32)
from ._lib import INSHipError
from .boundary_conditions import DirichletBC, HaloBC, PeriodicBC, PressureBC, SymmetricBC
from .distributed import AbiSlabComm, HipSlabKernels, SlabComm, SlabLayout, SlabStepper
from .grid import Grid, cosine_grid, max_size, stretched_grid, tanh_grid
from .initializers import random_field, temperaturefield, velocityfield
from .operators import (
    Dfield,
    Dfield_,
    Qfield,
    Qfield_,
    apply_bc_p,
    apply_bc_p_,
    apply_bc_temp,
    apply_bc_temp_,
    applybodyforce,
    applybodyforce_,
    convection_diffusion_temp,
    convection_diffusion_temp_,
    dissipation,
    dissipation_,
    dissipation_from_strain,
    dissipation_from_strain_,
    divoftensor_,
    eig2field,
    eig2field_,
    gravity,
    gravity_,
    interpolate_u_p,
    interpolate_u_p_,
    interpolate_ω_p,
    interpolate_ω_p_,
    smagorinsky_closure,
    smagtensor_,
    tensorbasis,
    tensorbasis_,
    tensorbasis_matrices,
    tensorfield,
    vorticity,
    vorticity_,
    apply_bc_u,
    apply_bc_u_,
    applypressure,
    applypressure_,
    convection,
    convection_,
    convectiondiffusion_,
    diffusion,
    diffusion_,
    divergence,
    divergence_,
    kinetic_energy,
    kinetic_energy_,
    laplacian,
    laplacian_,
    max_abs_divergence,
    momentum,
    momentum_,
    pressuregradient,
    pressuregradient_,
    scalewithvolume,
    scalewithvolume_,
    total_kinetic_energy,
)
from .pressure import (
    default_psolver,
    poisson,
    poisson_,
    pressure,
    project,
    project_,
    psolver_cg,
    psolver_direct,
    psolver_spectral,
)
from .processors import (Observable, fieldsaver, get_scale_numbers, observefield, observespectrum, processor, save_vtk, spectral_stuff, timelogger,
                         vtk_writer)
from .setup import Setup, copyfield, from_numpy, scalarfield, temperature_equation, to_numpy, vectorfield
from .sciml import create_right_hand_side, right_hand_side_
from .solver import get_cfl_timestep_, get_state, solve_unsteady
from .time_steppers import (
    ExplicitRungeKuttaMethod,
    LMWray3,
    RKMethods,
    create_stepper,
    ode_method_cache,
    runge_kutta_method,
    timestep,
    timestep_,
    timesteps_,
)

_lib.load()  # fail loudly at import time if libinship.so is absent
