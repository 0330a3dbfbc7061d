"""Host mirror of the `_f32` entry-point family (include/ins_hip.h; csrc/ins_f32.hip, csrc/ins_f32g.hip): the reference with `T = Float32`
(docs/src/manual/precision.md:3-16, examples/DecayingTurbulence3D.jl:16): all-periodic uniform boxes on the spectral solver, every other grid
(walls, symmetric / pressure sides, stretched spacings; constant boundary data) on `psolver_wrap32` around an fp64 solver.

Fields are torch.float32 tensors in the reference layout; `setup` is the ordinary (fp64-metric) Setup."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .setup import _fortran_strides


def _alloc32(setup, shape):
    t = torch.zeros(tuple(reversed(shape)), dtype=torch.float32, device=setup.device)
    return t.permute(*reversed(range(len(shape))))


def scalarfield32(setup):
    return _alloc32(setup, setup.grid.N)


def vectorfield32(setup):
    return _alloc32(setup, setup.grid.N + (setup.grid.dimension,))


def to_f32(setup, f):
    """Round a float64 field (torch, reference layout) or a numpy array of field shape to a float32 device field."""
    out = _alloc32(setup, tuple(f.shape))
    out.copy_(f if isinstance(f, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(f)).to(setup.device))
    return out


def _ptr(setup, f, ncomp):
    shape = setup.grid.N + ((ncomp,) if ncomp else ())
    if not isinstance(f, torch.Tensor) or f.dtype != torch.float32:
        raise TypeError("the _f32 family takes float32 torch tensors")
    if f.device != setup.device or tuple(f.shape) != shape or tuple(f.stride()) != _fortran_strides(shape):
        raise ValueError(f"field must live on {setup.device} with shape {shape} and column-major strides")
    return C.c_void_p(f.data_ptr())


def _constant_bc_only(setup):
    if setup.needs_bc_planes:
        raise NotImplementedError("the _f32 family takes constant boundary data (callable DirichletBC values: use the fp64 entry points)")


def apply_bc_u32_(u, setup):
    _constant_bc_only(setup)
    _lib.call("ins_apply_bc_u_f32", setup.handle, _ptr(setup, u, setup.grid.dimension), setup.stream)
    return u


def apply_bc_p32_(p, setup):
    _lib.call("ins_apply_bc_p_f32", setup.handle, _ptr(setup, p, 0), setup.stream)
    return p


def momentum32_(F, u, setup):
    """momentum!(F, u, nothing, t, setup) with T = Float32 (operators.jl:967-976)."""
    D = setup.grid.dimension
    _lib.call("ins_momentum_f32", setup.handle, float(1.0 / setup.Re), _ptr(setup, u, D), _ptr(setup, F, D), setup.stream)
    return F


class psolver_spectral32:
    """psolver_spectral(setup) with T = Float32 (pressure.jl:289-351)."""

    def __init__(self, setup):
        self.setup = setup
        self._handle = C.c_void_p()
        _lib.call("ins_poisson_spectral_create_f32", setup.handle, C.byref(self._handle))

    @property
    def handle(self):
        return self._handle

    def __call__(self, p):
        _lib.call("ins_poisson_solve_f32", self._handle, _ptr(self.setup, p, 0), self.setup.stream)
        return p

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            try:
                _lib.load().ins_poisson_destroy_f32(h)
            except Exception:
                pass


class psolver_wrap32(psolver_spectral32):
    """psolver_direct(setup) / psolver_cg(setup) / psolver_spectral(setup) with T = Float32 on any grid (pressure.jl:85-154, 209-351): a Float32 solver
    around the fp64 solver `psolver64` (default: the setup's default_psolver) — right-hand side in double, fp64 solve, pressure rounded once."""

    def __init__(self, setup, psolver64=None):
        from .pressure import default_psolver

        self.setup = setup
        self.psolver64 = psolver64 if psolver64 is not None else default_psolver(setup)  # kept alive: the native handle borrows it
        self._handle = C.c_void_p()
        _lib.call("ins_poisson_wrap_f32", setup.handle, self.psolver64.handle, C.byref(self._handle))


def default_psolver32(setup):
    """default_psolver(setup) with T = Float32 (pressure.jl:85-98): spectral on all-periodic uniform boxes, direct elsewhere."""
    from .pressure import default_psolver, psolver_spectral

    ps64 = default_psolver(setup)
    if isinstance(ps64, psolver_spectral):
        del ps64
        return psolver_spectral32(setup)
    return psolver_wrap32(setup, ps64)


def project32_(u, setup, psolver, p):
    """project!(u, setup; psolver, p) with T = Float32 (pressure.jl:69-82)."""
    D = setup.grid.dimension
    _lib.call("ins_project_f32", setup.handle, psolver.handle, _ptr(setup, u, D), _ptr(setup, p, 0), setup.stream)
    return u


def max_abs_divergence32(u, setup, psolver):
    out = C.c_float()
    _lib.call("ins_max_abs_divergence_f32", setup.handle, psolver.handle, _ptr(setup, u, setup.grid.dimension), C.byref(out), setup.stream)
    return float(out.value)


class ERKCache32:
    """ode_method_cache(method, setup) with T = Float32 (time_stepper_caches.jl:34-49)."""

    def __init__(self, method, setup, psolver):
        self.setup, self.psolver = setup, psolver
        A = np.ascontiguousarray(method.A, dtype=np.float64)
        c = np.ascontiguousarray(method.c, dtype=np.float64)
        self._handle = C.c_void_p()
        dp = C.POINTER(C.c_double)
        _constant_bc_only(setup)
        _lib.call("ins_rk_create_f32", setup.handle, psolver.handle, len(method.b), A.ctypes.data_as(dp), c.ctypes.data_as(dp), C.byref(self._handle))

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            try:
                _lib.load().ins_rk_destroy_f32(h)
            except Exception:
                pass


def timestep32_(cache, u, Δt):
    """One explicit RK step of the float32 field `u` in place (step_explicit_runge_kutta.jl:4-59)."""
    s = cache.setup
    _lib.call("ins_rk_step_f32", cache._handle, float(1.0 / s.Re), _ptr(s, u, s.grid.dimension), float(Δt), s.stream)
    return u


def timesteps32_(cache, u, Δt, nsteps):
    """`nsteps` explicit RK steps of size Δt of the float32 field `u` in place: the fixed-Δt loop of solve_unsteady (solver.jl:74-83) as one native call."""
    s = cache.setup
    _lib.call("ins_rk_steps_f32", cache._handle, float(1.0 / s.Re), _ptr(s, u, s.grid.dimension), float(Δt), int(nsteps), s.stream)
    return u
