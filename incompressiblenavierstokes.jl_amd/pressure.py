"""Pressure solvers and projection (pressure.jl), host side: psolver objects wrap `ins_poisson_t`."""
import ctypes as C
import math

import numpy as np

from . import _lib
from .boundary_conditions import PeriodicBC
from .operators import (
    apply_bc_p,
    apply_bc_u,
    divergence,
    momentum,
    pressuregradient,
    scalewithvolume,
)
from .setup import copyfield, scalarfield


class _PSolver:
    """A `psolver` closure `p -> p` (pressure.jl:143, 222, 318) backed by a libinship handle."""

    kind = "?"

    def __init__(self, setup):
        self.setup = setup
        self._handle = C.c_void_p()

    @property
    def handle(self):
        return self._handle

    def __call__(self, p):
        _lib.call("ins_poisson_solve_f64", self._handle, self.setup.ptr(p, False), self.setup.stream)
        return p

    def last_info(self):
        it, res = C.c_int64(), C.c_double()
        _lib.call("ins_poisson_last_info", self._handle, C.byref(it), C.byref(res))
        return it.value, res.value

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            try:
                _lib.load().ins_poisson_destroy(h)
            except Exception:
                pass


class psolver_spectral(_PSolver):
    """Create spectral Poisson solver from setup (pressure.jl:289-351): rocFFT + fused k-space kernel."""

    kind = "spectral"

    def __init__(self, setup):
        super().__init__(setup)
        _lib.call("ins_poisson_spectral_create", setup.handle, C.byref(self._handle))


class psolver_cg(_PSolver):
    """Conjugate gradients iterative Poisson solver (pressure.jl:209-286), Jacobi preconditioner."""

    kind = "cg"

    def __init__(self, setup, abstol=0.0, reltol=math.sqrt(np.finfo(np.float64).eps), maxiter=None, bordered=False):
        super().__init__(setup)
        _lib.call("ins_poisson_cg_create", setup.handle, float(abstol), float(reltol), int(maxiter or 0), C.byref(self._handle))
        if bordered:  # psolver_direct's treatment of the singular system (pressure.jl:133-140)
            _lib.call("ins_poisson_cg_bordered", self._handle, 1)


def psolver_direct(setup):
    """pressure.jl:101-154.  The reference's CPU method factorises `laplacian_mat` with SuiteSparse and
    its GPU method needs cuDSS (ext/IncompressibleNavierStokesCUDSSExt.jl); a device sparse direct solver
    is out of this round's scope — non-periodic problems use `psolver_cg`."""
    raise NotImplementedError(
        "psolver_direct has no MI355X implementation yet; use psolver_cg(setup) (tight reltol for direct-solver accuracy)"
    )


def default_psolver(setup):
    """Get default Poisson solver from setup (pressure.jl:85-98); the non-spectral branch returns
    `psolver_cg(reltol=1e-12, bordered=True)`: the same linear system the direct solver factorises."""
    g = setup.grid
    isperiodic = all(isinstance(a, PeriodicBC) and isinstance(b, PeriodicBC) for a, b in setup.boundary_conditions)
    isuniform = all(np.allclose(d, d[0], rtol=math.sqrt(np.finfo(np.float64).eps), atol=0) for d in g.Δ)
    if isperiodic and isuniform:
        return psolver_spectral(setup)
    return psolver_cg(setup, reltol=1e-12, bordered=True)


def poisson_(psolver, f):
    """Solve the Poisson equation for the pressure, in place (pressure.jl:22)."""
    return psolver(f)


def poisson(psolver, f):
    """pressure.jl:15"""
    return psolver(copyfield(f))


def project_(u, setup, psolver, p):
    """Project velocity field onto divergence-free space, in place (pressure.jl:69-82)."""
    _lib.call("ins_project_f64", setup.handle, psolver.handle, setup.ptr(u, True), setup.ptr(p, False), setup.stream)
    return u


def project(u, setup, psolver):
    """pressure.jl:52-66 (allocating twin, operator by operator as the reference)."""
    div = scalewithvolume(divergence(u, setup), setup)
    p = poisson(psolver, div)
    p = apply_bc_p(p, 0.0, setup)
    G = pressuregradient(p, setup)
    return u - G


def pressure(u, temp, t, setup, psolver):
    """Compute pressure from velocity field (pressure.jl:30-38)."""
    F = momentum(u, temp, t, setup)
    F = apply_bc_u(F, t, setup, dudt=True)
    div = scalewithvolume(divergence(F, setup), setup)
    p = poisson(psolver, div)
    return apply_bc_p(p, t, setup)
