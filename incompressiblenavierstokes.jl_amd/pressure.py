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
        _lib.sync_fft_plan_caches()


class psolver_cg(_PSolver):
    """Conjugate gradients iterative Poisson solver (pressure.jl:209-286), Jacobi preconditioner."""

    kind = "cg"

    def __init__(self, setup, abstol=0.0, reltol=math.sqrt(np.finfo(np.float64).eps), maxiter=None, bordered=False):
        super().__init__(setup)
        _lib.call("ins_poisson_cg_create", setup.handle, float(abstol), float(reltol), int(maxiter or 0), C.byref(self._handle))
        if bordered:  # psolver_direct's treatment of the singular system (pressure.jl:133-140)
            _lib.call("ins_poisson_cg_bordered", self._handle, 1)


def _laplacian_1d(setup, a):
    """The 1-D factor Tα of `laplacian!` (operators.jl:328-350) on Ip[α]: Ω/Δα · ((p₊-p)/Δu[I] - (p-p₋)/Δu[I-1]) with
    the reference's branch order for the first/last volume; ghost pressures of the `else` branch follow
    apply_bc_p! (boundary_conditions.jl:186-215: periodic wrap, Neumann copy for Dirichlet/Symmetric, 0 for Pressure)."""
    from .boundary_conditions import DirichletBC, PressureBC

    g = setup.grid
    lo, hi = g.Ip[a]
    n = hi - lo
    du = g.Δu[a]
    bl, br = setup.boundary_conditions[a]
    T = np.zeros((n, n))
    for i in range(n):
        cr, cl = 1.0 / du[lo + i], 1.0 / du[lo + i - 1]
        first, last = i == 0, i == n - 1
        right = left = True  # which one-sided differences survive
        ghost_r = ghost_l = None  # column the ghost value aliases, None = 0 (PressureBC)
        if first and isinstance(bl, PressureBC):
            pass
        elif last and isinstance(br, PressureBC):
            pass
        elif first and isinstance(bl, DirichletBC):
            left = False
        elif last and isinstance(br, DirichletBC):
            right = False
        else:
            if first:
                ghost_l = n - 1 if isinstance(bl, PeriodicBC) else i
            if last:
                ghost_r = 0 if isinstance(br, PeriodicBC) else i
        if right:
            T[i, i] -= cr
            j = i + 1 if not last else ghost_r
            if j is not None:
                T[i, j] += cr
        if left:
            T[i, i] -= cl
            j = i - 1 if not first else ghost_l
            if j is not None:
                T[i, j] += cl
    return T


def _sym_slack(S):
    """How far a decomposition may be from the given matrix and still count as 'at rounding level': 64 x the backward-error bound n·eps·max|S| of the
    symmetric eigensolver itself (cosine grids are mirror-symmetric only up to the rounding of their coordinates: 1.5e-12 relative in the wall cells at
    n = 256, 5e-12 at n = 1024 — inside this slack; a grid that is merely close to symmetric is not)."""
    return 64.0 * S.shape[0] * np.finfo(float).eps * np.abs(S).max()


def _centrosymmetric(S):
    """A symmetric grid (cosine, tanh, uniform walls) gives a factor that commutes with the reflection i -> n-1-i."""
    n = S.shape[0]
    return n % 2 == 0 and n >= 8 and np.abs(S - S[::-1, ::-1]).max() <= _sym_slack(S)


def _eigh_1d(S):
    """Eigenpairs of the symmetric 1-D factor: through the even / odd blocks when the grid is mirror-symmetric, and only if those pairs satisfy the GIVEN
    matrix to rounding level (the even / odd route diagonalises the symmetrised matrix: on a grid that is only approximately symmetric its pairs belong to
    a perturbed operator and the projection residual grows to that perturbation times the condition number — advisor finding, round 2); else LAPACK's."""
    if _centrosymmetric(S):
        lam, W = _eigh_even_odd(S)
        if np.abs(S @ W - W * lam[None, :]).max() <= _sym_slack(S):
            return lam, W
    return np.linalg.eigh(S)


def _eigh_even_odd(S):
    """Eigen-decomposition of a centrosymmetric symmetric matrix through its even and odd blocks: S (x; Jx) = (y; Jy) with y = (A + B J) x and
    S (x; -Jx) = (y; -Jy) with y = (A - B J) x, A = S[:h, :h], B = S[:h, h:].  Two half-size problems, and every eigenvector comes out EXACTLY even
    or odd — on strongly stretched grids the plain solver returns arbitrary mixtures inside the nearly degenerate wall-mode pairs.  The library runs a
    direction with such eigenvectors as two half-size GEMMs on the folded data (csrc/ins_fdm.hip)."""
    n = S.shape[0]
    h = n // 2
    Sm = 0.25 * (S + S.T + S[::-1, ::-1] + S[::-1, ::-1].T)  # exactly symmetric and centrosymmetric
    A, BJ = Sm[:h, :h], Sm[:h, h:][:, ::-1]
    le, We = np.linalg.eigh(A + BJ)
    lo_, Wo = np.linalg.eigh(A - BJ)
    W = np.empty((n, n))
    W[:h, :h], W[h:, :h] = We, We[::-1, :]
    W[:h, h:], W[h:, h:] = Wo, -Wo[::-1, :]
    W /= np.sqrt(2.0)
    return np.concatenate([le, lo_]), W


class psolver_direct(_PSolver):
    """Create direct Poisson solver from setup (pressure.jl:101-154).

    The reference factorises `laplacian_mat(setup)` (SuiteSparse on the CPU, cuDSS on CUDA) — bordered with the
    constant vector when no side is a PressureBC (pressure.jl:133-140).  On the tensor-product grids of this
    package that matrix is Σα Tα ⊗ (⊗β≠α Dβ), so the same system is solved directly by fast diagonalisation:
    the generalised eigenpairs Tα Vα = Dα Vα Λα are computed here once (host, O(N³) per direction, like the
    reference's factorisation step) and every solve is six fp64 GEMMs on the device (csrc/ins_fdm.hip)."""

    kind = "direct"

    def __init__(self, setup):
        super().__init__(setup)
        g = setup.grid
        D = g.dimension
        self._V, self._lam = [], []
        for a in range(D):
            lo, hi = g.Ip[a]
            T = _laplacian_1d(setup, a)
            if not np.allclose(T, T.T, rtol=1e-13, atol=0):
                raise ValueError("psolver_direct: the 1-D Laplacian factor is not symmetric")
            dm = 1.0 / np.sqrt(g.Δ[a][lo:hi])
            S = dm[:, None] * T * dm[None, :]  # D^-1/2 T D^-1/2 = W Λ Wᵀ
            lam, W = _eigh_1d(S)
            self._V.append(np.asfortranarray(dm[:, None] * W))  # Vα = D^-1/2 W,  VαᵀDαVα = I
            self._lam.append(np.ascontiguousarray(lam))
        dp = C.POINTER(C.c_double)
        Vp = (dp * 3)(*[v.ctypes.data_as(dp) for v in self._V])
        lp = (dp * 3)(*[v.ctypes.data_as(dp) for v in self._lam])
        _lib.call("ins_poisson_fdm_create", setup.handle, Vp, lp, C.byref(self._handle))


def default_psolver(setup):
    """Get default Poisson solver from setup (pressure.jl:85-98)."""
    g = setup.grid
    isperiodic = all(isinstance(a, PeriodicBC) and isinstance(b, PeriodicBC) for a, b in setup.boundary_conditions)
    isuniform = all(np.allclose(d, d[0], rtol=math.sqrt(np.finfo(np.float64).eps), atol=0) for d in g.Δ)
    if isperiodic and isuniform:
        return psolver_spectral(setup)
    return psolver_direct(setup)


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
