"""
CPU restatement (numpy, fp64) of the IncompressibleNavierStokes.jl hot path.

*** TEST INFRASTRUCTURE ONLY. ***  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module, and only as the checker.
The product (`incompressiblenavierstokes.jl_amd/`) never imports it and has no CPU fallback.

How parity is pinned: the reference is 100 % Julia, Julia is absent from the build container, and
the reference ships no golden vectors.  This restatement is therefore pinned by
the reference's OWN known-answer tests and invariants, re-run against it in `tests/test_oracle_*.py`:
  test/psolvers.jl:1-32 (analytic Poisson solution, direct/cg/spectral), test/operators.jl:58-160
  (G = -D', laplacian! == laplacian_mat, skew-symmetric convection, dissipative diffusion,
  fused == unfused), test/matrices.jl:19-51 (BC / divergence / gradient / diffusion matrices on the mixed-BC
  fixture: the matrices are restated by INDEX ASSEMBLY in oracle/ins_matrices.py — src/matrices.jl:1-555, the
  reference's second implementation, which never calls the stencil code below — tests/test_oracle_matrices.py),
  test/timesteppers.jl:1-43, examples/TaylorGreenVortex2D.jl:29-74 (analytic decay, n^-2 convergence).

Conventions.  All indices here are 0-based; the reference is 1-based.  Fields are Fortran-ordered
numpy arrays of shape `N + (D,)` (vector) / `N` (scalar) so that `u[i, j, k, a]` addresses the same
memory cell as Julia's `u[i+1, j+1, k+1, a+1]` (initializers.jl:2-6).  Index ranges are python
`(lo, hi)` half-open pairs:  Julia `a:b`  ==  `(a-1, b)`.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Optional, Sequence, Tuple, Union

import numpy as np

EPS = float(np.finfo(np.float64).eps)


# --------------------------------------------------------------------------------------
# Boundary-condition types                                  boundary_conditions.jl:2-36
# --------------------------------------------------------------------------------------
class AbstractBC:
    pass


class PeriodicBC(AbstractBC):
    def __repr__(self):
        return "PeriodicBC()"


class DirichletBC(AbstractBC):
    """`u` is None (no slip), a tuple of constants, or a callable (alpha, x..., t)."""

    def __init__(self, u=None):
        self.u = u

    def __repr__(self):
        return f"DirichletBC({self.u!r})"


class SymmetricBC(AbstractBC):
    def __repr__(self):
        return "SymmetricBC()"


class PressureBC(AbstractBC):
    def __repr__(self):
        return "PressureBC()"


def padghost(bc, x: list, isright: bool) -> None:
    """boundary_conditions.jl:42-61.  `x` is a python list, modified in place."""
    if isinstance(bc, PeriodicBC):
        if isright:
            x.append(x[-1] + (x[1] - x[0]))
        else:
            x.insert(0, x[0] - (x[-1] - x[-2]))
    elif isinstance(bc, DirichletBC):
        if isright:
            x.append(x[-1])
        else:
            x.insert(0, x[0])
    elif isinstance(bc, SymmetricBC):
        if isright:
            x.append(x[-1] + (x[-1] - x[-2]))
        else:
            x.insert(0, x[0] - (x[1] - x[0]))
    elif isinstance(bc, PressureBC):
        if isright:
            x.append(x[-1])
        else:
            x.insert(0, x[0])
            x.insert(0, x[0])
    else:
        raise TypeError(bc)


def offset_u(bc, isright: bool, isnormal: bool) -> int:
    """boundary_conditions.jl:79-89 (argument order as in the methods, not the docstring)."""
    if isinstance(bc, PeriodicBC):
        return 1
    if isinstance(bc, (DirichletBC, SymmetricBC)):
        return 1 + int(isright and isnormal)
    if isinstance(bc, PressureBC):
        return 1 + int((not isright) and (not isnormal))
    raise TypeError(bc)


def offset_p(bc, isright: bool) -> int:
    if isinstance(bc, (PeriodicBC, DirichletBC, SymmetricBC)):
        return 1
    if isinstance(bc, PressureBC):
        return 1 + int(not isright)
    raise TypeError(bc)


# --------------------------------------------------------------------------------------
# Grid generators                                                        grid.jl:39-77
# --------------------------------------------------------------------------------------
def cosine_grid(a, b, N):
    i = np.arange(N + 1, dtype=np.float64)
    return a + (b - a) * (1 - np.cos(np.pi * (i / N))) / 2


def stretched_grid(a, b, N, s=1.0):
    if s <= 0:
        raise ValueError("The stretch factor must be positive")
    if np.isclose(s, 1.0):
        return np.linspace(a, b, N + 1)
    i = np.arange(N + 1, dtype=np.float64)
    return a + (b - a) * (1 - s**i) / (1 - s**N)


def tanh_grid(a, b, N, gamma=1.0):
    x = np.linspace(0.0, 1.0, N + 1)
    return a + (b - a) * (1 + np.tanh(gamma * (2 * x - 1)) / np.tanh(gamma)) / 2


# --------------------------------------------------------------------------------------
# Grid                                                                  grid.jl:100-276
# --------------------------------------------------------------------------------------
@dataclass
class Grid:
    D: int
    xlims: tuple
    N: tuple
    Nu: tuple
    Np: tuple
    Iu: tuple  # Iu[a][b] = (lo, hi)
    Ip: tuple  # Ip[b]    = (lo, hi)
    x: tuple
    xu: tuple
    xp: tuple
    dx: tuple  # Δ
    dxu: tuple  # Δu
    A: tuple  # A[a][b] = (A1, A2)


def make_grid(x: Sequence, boundary_conditions) -> Grid:
    xs = [list(map(float, np.asarray(xi, dtype=np.float64))) for xi in x]
    xlims = tuple((min(xi), max(xi)) for xi in xs)
    D = len(xs)
    for d in range(D):
        a, b = boundary_conditions[d]
        padghost(a, xs[d], False)
        padghost(b, xs[d], True)
    xs = [np.array(xi, dtype=np.float64) for xi in xs]
    N = tuple(len(xi) - 1 for xi in xs)
    bcs = boundary_conditions

    def rng_u(al, be):
        na = offset_u(bcs[be][0], False, al == be)
        nb = offset_u(bcs[be][1], True, al == be)
        return (na, N[be] - nb)

    def rng_p(be):
        na = offset_p(bcs[be][0], False)
        nb = offset_p(bcs[be][1], True)
        return (na, N[be] - nb)

    Iu = tuple(tuple(rng_u(al, be) for be in range(D)) for al in range(D))
    Ip = tuple(rng_p(be) for be in range(D))
    Nu = tuple(tuple(hi - lo for (lo, hi) in Iu[al]) for al in range(D))
    Np = tuple(hi - lo for (lo, hi) in Ip)
    xp = tuple((xi[:-1] + xi[1:]) / 2 for xi in xs)
    xu = tuple(tuple(xs[be][1:] if al == be else xp[be] for be in range(D)) for al in range(D))
    dx = tuple(np.maximum(np.diff(xi), EPS) for xi in xs)
    dxu = tuple(
        np.maximum(np.concatenate([np.diff(xp[d]), [dx[d][-1] / 2]]), EPS) for d in range(D)
    )

    def weights(al, be):
        if al == be:
            A1 = np.full(N[al], 0.5)
            A1[0] = 1.0
            A2 = np.full(N[al], 0.5)
            A2[-1] = 1.0
        else:
            A2i = (xs[be][1 : N[be]] - xp[be][: N[be] - 1]) / dxu[be][: N[be] - 1]
            A1 = np.concatenate([[1.0], 1.0 - A2i])
            A2 = np.concatenate([A2i, [1.0]])
        return (A1, A2)

    A = tuple(tuple(weights(al, be) for be in range(D)) for al in range(D))
    return Grid(D, xlims, N, Nu, Np, Iu, Ip, tuple(xs), xu, xp, dx, dxu, A)


# --------------------------------------------------------------------------------------
# Setup                                                                   setup.jl:2-46
# --------------------------------------------------------------------------------------
@dataclass
class Setup:
    grid: Grid
    boundary_conditions: tuple
    Re: float
    bodyforce: object = None
    closure_model: object = None
    temperature: object = None
    issteadybodyforce: bool = True


def make_setup(x, boundary_conditions=None, Re=1000.0) -> Setup:
    D = len(x)
    if boundary_conditions is None:
        boundary_conditions = tuple((PeriodicBC(), PeriodicBC()) for _ in range(D))
    return Setup(make_grid(x, boundary_conditions), tuple(boundary_conditions), float(Re))


def scalarfield(setup) -> np.ndarray:  # initializers.jl:2
    return np.zeros(setup.grid.N, dtype=np.float64, order="F")


def vectorfield(setup) -> np.ndarray:  # initializers.jl:5-6
    return np.zeros(setup.grid.N + (setup.grid.D,), dtype=np.float64, order="F")


# ---------------------------------------------------------------------------- helpers
def _sl(ranges, shift=None):
    """Tuple of slices for half-open per-dimension ranges, optionally shifted."""
    D = len(ranges)
    shift = shift or (0,) * D
    return tuple(slice(lo + s, hi + s) for (lo, hi), s in zip(ranges, shift))


def _e(D, a, s=1):
    return tuple(s if b == a else 0 for b in range(D))


def _add(a, b):
    return tuple(x + y for x, y in zip(a, b))


def _vec(v, rng, axis, D, shift=0):
    """1-D metric `v[lo+shift:hi+shift]` reshaped to broadcast along `axis` of a D-dim block."""
    lo, hi = rng
    shape = [1] * D
    shape[axis] = hi - lo
    return v[lo + shift : hi + shift].reshape(shape)


def _inner(setup, rng):
    """Intersect ranges with the kernels' `ndrange = N .- 2`, offset 1 (operators.jl:166,385)."""
    N = setup.grid.N
    return tuple((max(lo, 1), min(hi, n - 1)) for (lo, hi), n in zip(rng, N))


# --------------------------------------------------------------------------------------
# Ghost fill                              boundary_conditions.jl:97-103, 159-206, 276-502
# --------------------------------------------------------------------------------------
def _plane(N, be, i):
    return tuple(slice(i, i + 1) if b == be else slice(0, N[b]) for b in range(len(N)))


def apply_bc_u_(u, t, setup, dudt=False):
    g = setup.grid
    D, N = g.D, g.N
    for be in range(D):
        for isright in (False, True):
            bc = setup.boundary_conditions[be][int(isright)]
            if isinstance(bc, PeriodicBC):
                if isright:
                    continue
                ia, ib = g.Ip[be][0] - 1, g.Ip[be][1]
                u[_plane(N, be, ia)] = u[_plane(N, be, ib - 1)]
                u[_plane(N, be, ib)] = u[_plane(N, be, ia + 1)]
            elif isinstance(bc, DirichletBC):
                for al in range(D):
                    lo, hi = g.Iu[al][be]
                    i = hi if isright else lo - 1
                    I = _plane(N, be, i)
                    if bc.u is None:
                        u[I + (al,)] = 0.0
                    elif isinstance(bc.u, tuple):
                        u[I + (al,)] = 0.0 if dudt else float(bc.u[al])
                    else:
                        xs = []
                        for ga in range(D):
                            coords = g.xu[al][ga][I[ga]]
                            shape = [1] * D
                            shape[ga] = coords.size
                            xs.append(coords.reshape(shape))
                        if dudt:
                            h = math.sqrt(EPS) / 2
                            val = (bc.u(al, *xs, t + h) - bc.u(al, *xs, t - h)) / (2 * h)
                        else:
                            val = bc.u(al, *xs, t)
                        u[I + (al,)] = np.broadcast_to(val, u[I + (al,)].shape)
            elif isinstance(bc, SymmetricBC):
                for al in range(D):
                    lo, hi = g.Iu[al][be]
                    i = hi if isright else lo - 1
                    if al == be:
                        u[_plane(N, be, i) + (al,)] = 0.0
                    else:
                        j = i - 1 if isright else i + 1
                        u[_plane(N, be, i) + (al,)] = u[_plane(N, be, j) + (al,)]
            elif isinstance(bc, PressureBC):
                for al in range(D):
                    lo, hi = g.Iu[al][be]
                    i = hi if isright else lo - 1
                    j = i - 1 if isright else i + 1
                    u[_plane(N, be, i) + (al,)] = u[_plane(N, be, j) + (al,)]
            else:
                raise TypeError(bc)
    return u


def apply_bc_p_(p, t, setup):
    g = setup.grid
    D, N = g.D, g.N
    for be in range(D):
        for isright in (False, True):
            bc = setup.boundary_conditions[be][int(isright)]
            lo, hi = g.Ip[be]
            i = hi if isright else lo - 1
            if isinstance(bc, PeriodicBC):
                if isright:
                    continue
                p[_plane(N, be, lo - 1)] = p[_plane(N, be, hi - 1)]
                p[_plane(N, be, hi)] = p[_plane(N, be, lo)]
            elif isinstance(bc, DirichletBC):
                pass  # boundary_conditions.jl:388
            elif isinstance(bc, SymmetricBC):
                j = i - 1 if isright else i + 1
                p[_plane(N, be, i)] = p[_plane(N, be, j)]
            elif isinstance(bc, PressureBC):
                p[_plane(N, be, i)] = 0.0
            else:
                raise TypeError(bc)
    return p


def apply_bc_u(u, t, setup, **kw):
    return apply_bc_u_(u.copy(order="F"), t, setup, **kw)


def apply_bc_p(p, t, setup):
    return apply_bc_p_(p.copy(order="F"), t, setup)


# --------------------------------------------------------------------------------------
# Operators                                                               operators.jl
# --------------------------------------------------------------------------------------
def scalewithvolume_(p, setup):  # operators.jl:81-95 (whole padded array)
    g = setup.grid
    for a in range(g.D):
        shape = [1] * g.D
        shape[a] = g.N[a]
        p *= g.dx[a].reshape(shape)
    return p


def scalewithvolume(p, setup):
    return scalewithvolume_(p.copy(order="F"), setup)


def divergence_(div, u, setup):  # operators.jl:106-125
    g = setup.grid
    D = g.D
    R = g.Ip
    d = np.zeros(tuple(hi - lo for lo, hi in R))
    for a in range(D):
        d = d + (u[_sl(R) + (a,)] - u[_sl(R, _e(D, a, -1)) + (a,)]) / _vec(g.dx[a], R[a], a, D)
    div[_sl(R)] = d
    return div


def divergence(u, setup):
    return divergence_(scalarfield(setup), u, setup)


def pressuregradient_(G, p, setup):  # operators.jl:159-178
    g = setup.grid
    D = g.D
    for a in range(D):
        R = _inner(setup, g.Iu[a])
        G[_sl(R) + (a,)] = (p[_sl(R, _e(D, a))] - p[_sl(R)]) / _vec(g.dxu[a], R[a], a, D)
    return G


def pressuregradient(p, setup):
    return pressuregradient_(vectorfield(setup), p, setup)


def applypressure_(u, p, setup):  # operators.jl:214-233
    g = setup.grid
    D = g.D
    for a in range(D):
        R = _inner(setup, g.Iu[a])
        u[_sl(R) + (a,)] -= (p[_sl(R, _e(D, a))] - p[_sl(R)]) / _vec(g.dxu[a], R[a], a, D)
    return u


def laplacian_(L, p, setup):  # operators.jl:297-364
    g = setup.grid
    D = g.D
    R = g.Ip
    L[...] = 0.0
    om = np.ones(tuple(hi - lo for lo, hi in R))
    for a in range(D):
        om = om * _vec(g.dx[a], R[a], a, D)
    for a in range(D):
        bc = setup.boundary_conditions[a]
        lo, hi = R[a]
        pc = p[_sl(R)]
        pp = p[_sl(R, _e(D, a))]
        pm = p[_sl(R, _e(D, a, -1))]
        da = _vec(g.dx[a], R[a], a, D)
        dur = _vec(g.dxu[a], R[a], a, D)
        dul = _vec(g.dxu[a], R[a], a, D, shift=-1)
        right = (pp - pc) / dur
        left = (pc - pm) / dul
        idx = np.arange(lo, hi).reshape([-1 if b == a else 1 for b in range(D)])
        isfirst = idx == lo
        islast = idx == hi - 1
        # Branch order as in the reference's if/elseif chain (operators.jl:334-350)
        c1 = isfirst & isinstance(bc[0], PressureBC)
        c2 = islast & isinstance(bc[1], PressureBC) & ~c1
        c3 = isfirst & isinstance(bc[0], DirichletBC) & ~c1 & ~c2
        c4 = islast & isinstance(bc[1], DirichletBC) & ~c1 & ~c2 & ~c3
        right = np.where(c2, (-pc) / dur, right)
        right = np.where(c4, 0.0, right)
        left = np.where(c1, pc / dul, left)
        left = np.where(c3, 0.0, left)
        L[_sl(R)] += om / da * (right - left)
    return L


def laplacian(p, setup):
    return laplacian_(scalarfield(setup), p, setup)


def _conv_terms(u, setup, a, b, R):
    """uαβ1, uαβ2, uβα1, uβα2 of operators.jl:404-409 / 670-675 on block R."""
    g = setup.grid
    D = g.D
    ea, eb = _e(D, a), _e(D, b)
    meb = _e(D, b, -1)
    A1, A2 = g.A[b][a]  # weights of component b in direction a  (reverse interpolation)
    uab1 = (u[_sl(R, meb) + (a,)] + u[_sl(R) + (a,)]) / 2
    uab2 = (u[_sl(R) + (a,)] + u[_sl(R, eb) + (a,)]) / 2
    s2 = -1 if a == b else 0
    s1 = 0 if a == b else 1
    uba1 = _vec(A2, R[a], a, D, shift=s2) * u[_sl(R, meb) + (b,)] + _vec(
        A1, R[a], a, D, shift=s1
    ) * u[_sl(R, _add(meb, ea)) + (b,)]
    uba2 = _vec(A2, R[a], a, D) * u[_sl(R) + (b,)] + _vec(A1, R[a], a, D, shift=1) * u[
        _sl(R, ea) + (b,)
    ]
    return uab1, uab2, uba1, uba2


def _diff_terms(u, setup, a, b, R):
    """∂βuα1, ∂βuα2 with the Δ > 2eps masks (operators.jl:559-567 / 668-684)."""
    g = setup.grid
    D = g.D
    eb, meb = _e(D, b), _e(D, b, -1)
    if a == b:
        da = _vec(g.dx[b], R[b], b, D)
        db = _vec(g.dx[b], R[b], b, D, shift=1)
    else:
        da = _vec(g.dxu[b], R[b], b, D, shift=-1)
        db = _vec(g.dxu[b], R[b], b, D)
    with np.errstate(over="ignore", invalid="ignore"):
        d1 = (u[_sl(R) + (a,)] - u[_sl(R, meb) + (a,)]) / da
        d2 = (u[_sl(R, eb) + (a,)] - u[_sl(R) + (a,)]) / db
    # Julia `false * x` is a strong zero
    d1 = np.where(da > 2 * EPS, d1, 0.0)
    d2 = np.where(db > 2 * EPS, d2, 0.0)
    return d1, d2


def convection_(F, u, setup):  # operators.jl:378-415   (adds to F)
    g = setup.grid
    D = g.D
    for a in range(D):
        R = _inner(setup, g.Iu[a])
        f = F[_sl(R) + (a,)].copy()
        for b in range(D):
            dab = _vec(g.dxu[b] if a == b else g.dx[b], R[b], b, D)
            uab1, uab2, uba1, uba2 = _conv_terms(u, setup, a, b, R)
            f = f - (uab2 * uba2 - uab1 * uba1) / dab
        F[_sl(R) + (a,)] = f
    return F


def convection(u, setup):
    return convection_(vectorfield(setup), u, setup)


def diffusion_(F, u, setup, use_viscosity=True):  # operators.jl:537-573   (adds to F)
    g = setup.grid
    D = g.D
    visc = 1.0 / setup.Re if use_viscosity else 1.0
    for a in range(D):
        R = _inner(setup, g.Iu[a])
        f = F[_sl(R) + (a,)].copy()
        for b in range(D):
            dab = _vec(g.dxu[b] if a == b else g.dx[b], R[b], b, D)
            d1, d2 = _diff_terms(u, setup, a, b, R)
            f = f + visc * (d2 - d1) / dab
        F[_sl(R) + (a,)] = f
    return F


def diffusion(u, setup, use_viscosity=True):
    return diffusion_(vectorfield(setup), u, setup, use_viscosity)


def convectiondiffusion_(F, u, setup):  # operators.jl:634-690   (adds to F)
    g = setup.grid
    D = g.D
    visc = 1.0 / setup.Re
    for a in range(D):
        R = _inner(setup, g.Iu[a])
        f = F[_sl(R) + (a,)].copy()
        for b in range(D):
            dab = _vec(g.dxu[b] if a == b else g.dx[b], R[b], b, D)
            uab1, uab2, uba1, uba2 = _conv_terms(u, setup, a, b, R)
            d1, d2 = _diff_terms(u, setup, a, b, R)
            f = f + (visc * (d2 - d1) - (uab2 * uba2 - uab1 * uba1)) / dab
        F[_sl(R) + (a,)] = f
    return F


def momentum_(F, u, temp, t, setup):  # operators.jl:967-976
    assert temp is None and setup.bodyforce is None
    F[...] = 0.0
    convectiondiffusion_(F, u, setup)
    return F


def momentum(u, temp, t, setup):
    return momentum_(vectorfield(setup), u, temp, t, setup)


def kinetic_energy_(ke, u, setup, interpolate_first=False):  # operators.jl:1516-1545
    g = setup.grid
    D = g.D
    R = g.Ip
    k = np.zeros(tuple(hi - lo for lo, hi in R))
    for a in range(D):
        up = u[_sl(R) + (a,)]
        um = u[_sl(R, _e(D, a, -1)) + (a,)]
        k = k + ((up + um) ** 2 if interpolate_first else up**2 + um**2)
    ke[_sl(R)] = k / (8 if interpolate_first else 4)
    return ke


def total_kinetic_energy(u, setup, **kw):  # operators.jl:1551-1556
    k = kinetic_energy_(scalarfield(setup), u, setup, **kw)
    k = scalewithvolume(k, setup)
    return float(np.sum(k[_sl(setup.grid.Ip)]))


# --------------------------------------------------------------------------------------
# Matrix-free twin of `laplacian_mat = P' Ω M Bu G Bp P`               matrices.jl:484-492
# (Bu/Bp are the *linear* parts of the ghost fills: Dirichlet values -> 0.)
# --------------------------------------------------------------------------------------
def _homogeneous(setup):
    bcs = tuple(
        tuple(DirichletBC() if isinstance(b, DirichletBC) else b for b in side)
        for side in setup.boundary_conditions
    )
    return Setup(setup.grid, bcs, setup.Re)


def laplacian_mat_apply(pdof, setup):
    g = setup.grid
    hs = _homogeneous(setup)
    p = scalarfield(setup)
    p[_sl(g.Ip)] = np.asarray(pdof).reshape(g.Np, order="F")
    apply_bc_p_(p, 0.0, hs)
    G = pressuregradient(p, hs)
    apply_bc_u_(G, 0.0, hs)
    M = divergence(G, hs)
    scalewithvolume_(M, hs)
    return M[_sl(g.Ip)].reshape(-1, order="F")


def laplacian_mat(setup, dense=False):
    """`laplacian_mat` (matrices.jl:483-492) = P' Ω M Bu G Bp P from the index-assembled sparse factors of oracle/ins_matrices.py (the
    reference's own construction; independent of the stencil code in this file).  `laplacian_mat_probed` below is the third cross-check."""
    from . import ins_matrices

    L = ins_matrices.laplacian_mat(setup)
    return L.toarray() if dense else L


def laplacian_mat_probed(setup, dense=False):
    """`laplacian_mat` by coloured probing of the matrix-free definition above (this file's stencil operators): the operator couples a DOF only with its 2·D face neighbours, so DOFs whose
    indices differ by a multiple of `c >= 3` in every direction (with `c | Np` when periodic) can be
    probed together.  Returns scipy CSR (or a dense array)."""
    import itertools

    import scipy.sparse as sp

    g = setup.grid
    D, Np = g.D, g.Np
    n = int(np.prod(Np))
    strides = []
    for a in range(D):
        periodic = isinstance(setup.boundary_conditions[a][0], PeriodicBC)
        c = 3
        if periodic:
            while c < Np[a] and Np[a] % c != 0:
                c += 1
        strides.append(min(c, Np[a]))
    idx = np.indices(Np)  # idx[a][I] = I_a
    lin = np.arange(n).reshape(Np, order="F")
    rows, cols, vals = [], [], []
    for color in itertools.product(*[range(c) for c in strides]):
        mask = np.ones(Np, dtype=bool)
        for a in range(D):
            mask &= (idx[a] % strides[a]) == color[a]
        e = mask.astype(np.float64).reshape(-1, order="F")
        resp = laplacian_mat_apply(e, setup).reshape(Np, order="F")
        # attribute each response entry to the probed DOF within distance 1 (with periodic wrap)
        for off in [(0,) * D] + [_e(D, a, s) for a in range(D) for s in (-1, 1)]:
            src = [idx[a] - off[a] for a in range(D)]  # candidate source = I - off
            ok = np.ones(Np, dtype=bool)
            for a in range(D):
                periodic = isinstance(setup.boundary_conditions[a][0], PeriodicBC)
                if periodic:
                    src[a] = src[a] % Np[a]
                else:
                    ok &= (src[a] >= 0) & (src[a] < Np[a])
                    src[a] = np.clip(src[a], 0, Np[a] - 1)
            ok &= mask[tuple(src)] & (resp != 0)
            if any(off) and not all(Np[a] > 2 for a in range(D) if off[a]):
                # tiny periodic dims: +1 and -1 neighbours coincide; handled by the dense path
                raise ValueError("grid too small for coloured probing")
            r = lin[ok]
            c_ = lin[tuple(s_[ok] for s_ in src)]
            rows.append(r)
            cols.append(c_)
            vals.append(resp[ok])
    L = sp.coo_matrix(
        (np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)
    ).tocsr()
    # entries reached through two different offsets were each given the full response: average out
    cnt = sp.coo_matrix(
        (np.ones(sum(len(r) for r in rows)), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)
    ).tocsr()
    L.data /= cnt.data
    return L.toarray() if dense else L


def laplacian_mat_dense_probe(setup):
    """Column-by-column probing; the unambiguous (slow) definition, for tiny grids."""
    n = int(np.prod(setup.grid.Np))
    L = np.zeros((n, n))
    e = np.zeros(n)
    for j in range(n):
        e[j] = 1.0
        L[:, j] = laplacian_mat_apply(e, setup)
        e[j] = 0.0
    return L


# --------------------------------------------------------------------------------------
# Pressure solvers / projection                                            pressure.jl
# --------------------------------------------------------------------------------------
def assert_uniform_periodic(setup, what):  # utils.jl:1-13
    g = setup.grid
    assert all(
        isinstance(a, PeriodicBC) and isinstance(b, PeriodicBC) for a, b in setup.boundary_conditions
    ), what + " requires periodic boundary conditions."
    assert all(np.allclose(d, d[0]) for d in g.dx), what + " requires uniform grid spacing."
    assert all(n % 2 == 0 for n in g.N), what + " requires even number of volumes."


def spectral_symbols(setup):
    """ahat[a][k] = 4 Ω sin²(π k / Np[a]) / Δx[a]²                    pressure.jl:301-311"""
    g = setup.grid
    D = g.D
    dx0 = [float(d[0]) for d in g.dx]
    om = float(np.prod(dx0))
    kmax = [g.Np[a] // 2 + 1 if a == 0 else g.Np[a] for a in range(D)]
    return [
        4 * om * np.sin(np.pi * (np.arange(kmax[a]) / g.Np[a])) ** 2 / dx0[a] ** 2 for a in range(D)
    ]


def psolver_spectral(setup):  # pressure.jl:289-351
    assert_uniform_periodic(setup, "Spectral psolver")
    g = setup.grid
    D = g.D
    ahat = spectral_symbols(setup)
    axes = tuple(range(D - 1, -1, -1))  # real (halved) transform over axis 0 = x, as rfft does
    den = np.zeros([len(a) for a in ahat])
    for a in range(D):
        shape = [1] * D
        shape[a] = len(ahat[a])
        den = den + ahat[a].reshape(shape)

    def psolve_(p):
        pI = np.array(p[_sl(g.Ip)])
        phat = np.fft.rfftn(pI, axes=axes)
        with np.errstate(divide="ignore", invalid="ignore"):
            phat = -phat / den
        phat[(0,) * D] = 0.0
        pI = np.fft.irfftn(phat, s=[g.Np[a] for a in axes], axes=axes)
        p[_sl(g.Ip)] = pI
        return p

    return psolve_


def laplace_diag(setup):
    """Jacobi diagonal d of pressure.jl:191-201 on Ip."""
    g = setup.grid
    D = g.D
    R = g.Ip
    om = np.ones(tuple(hi - lo for lo, hi in R))
    for a in range(D):
        om = om * _vec(g.dx[a], R[a], a, D)
    d = np.zeros_like(om)
    for a in range(D):
        d = d - om / _vec(g.dx[a], R[a], a, D) * (
            1 / _vec(g.dxu[a], R[a], a, D) + 1 / _vec(g.dxu[a], R[a], a, D, shift=-1)
        )
    return d


def psolver_cg(setup, abstol=0.0, reltol=math.sqrt(EPS), maxiter=None, info=None):
    """pressure.jl:209-286 (Jacobi-preconditioned CG, zero initial guess)."""
    g = setup.grid
    Ip = _sl(g.Ip)
    maxiter = int(np.prod(g.Np)) if maxiter is None else maxiter
    dinv = 1.0 / laplace_diag(setup)

    def psolve_(p):
        r, L, q = scalarfield(setup), scalarfield(setup), scalarfield(setup)
        laplacian_(L, q, setup)
        r[...] = p - L
        rho_prev = 1.0
        residual = math.sqrt(float(np.sum(r[Ip] ** 2)))
        tolerance = max(reltol * residual, abstol)
        it = 0
        p[...] = 0.0
        while it < maxiter and residual > tolerance:
            L[Ip] = -r[Ip] * dinv  # preconditioner(L, r): z = -r/d
            rho = float(np.sum(L[Ip] * r[Ip]))
            beta = rho / rho_prev
            q[...] = L + beta * q
            apply_bc_p_(q, 0.0, setup)
            laplacian_(L, q, setup)
            alpha = rho / float(np.sum(q[Ip] * L[Ip]))
            p += alpha * q
            r -= alpha * L
            rho_prev = rho
            residual = math.sqrt(float(np.sum(r[Ip] ** 2)))
            it += 1
        if info is not None:
            info["iterations"] = it
            info["residual"] = residual
        return p

    return psolve_


def psolver_direct(setup):
    """pressure.jl:117-154: factorise `laplacian_mat`; bordered with a ones row/column when singular
    (sparse LU here in place of CHOLMOD's LDLt — same linear system)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    g = setup.grid
    L = laplacian_mat(setup)
    n = L.shape[0]
    isdefinite = any(
        isinstance(a, PressureBC) or isinstance(b, PressureBC) for a, b in setup.boundary_conditions
    )
    if not isdefinite:
        assert abs(L - L.T).max() < math.sqrt(EPS), "Matrix not symmetric"
        L = (L + L.T) / 2
        e = sp.csr_matrix(np.ones((n, 1)))
        Lb = sp.bmat([[L, e], [e.T, None]], format="csc")
    else:
        Lb = L.tocsc()
    fact = spla.splu(Lb)

    def psolve_(p):
        f = np.zeros(Lb.shape[0])
        f[:n] = p[_sl(g.Ip)].reshape(-1, order="F")
        sol = fact.solve(f)
        p[_sl(g.Ip)] = sol[:n].reshape(g.Np, order="F")
        return p

    return psolve_


def default_psolver(setup):  # pressure.jl:85-98
    g = setup.grid
    isperiodic = all(
        isinstance(a, PeriodicBC) and isinstance(b, PeriodicBC) for a, b in setup.boundary_conditions
    )
    isuniform = all(np.allclose(d, d[0]) for d in g.dx)
    return psolver_spectral(setup) if (isperiodic and isuniform) else psolver_direct(setup)


def poisson_(psolver, f):  # pressure.jl:22
    return psolver(f)


def poisson(psolver, f):
    return psolver(f.copy(order="F"))


def project_(u, setup, psolver, p):  # pressure.jl:69-82
    divergence_(p, u, setup)
    scalewithvolume_(p, setup)
    poisson_(psolver, p)
    apply_bc_p_(p, 0.0, setup)
    applypressure_(u, p, setup)
    return u


def project(u, setup, psolver):  # pressure.jl:52-66
    div = divergence(u, setup)
    div = scalewithvolume(div, setup)
    p = poisson(psolver, div)
    p = apply_bc_p(p, 0.0, setup)
    G = pressuregradient(p, setup)
    return u - G


# --------------------------------------------------------------------------------------
# Explicit Runge-Kutta                     methods.jl:184-240, RKMethods.jl, step_explicit_runge_kutta.jl
# --------------------------------------------------------------------------------------
@dataclass
class ExplicitRungeKuttaMethod:
    A: np.ndarray
    b: np.ndarray
    c: np.ndarray
    r: float = 0.0


def runge_kutta_method(A, b, c, r=0.0):
    A = np.array(A, dtype=np.float64)
    b = np.array(b, dtype=np.float64)
    c = np.array(c, dtype=np.float64)
    assert np.allclose(np.triu(A), 0), "only explicit tableaux are on the path"
    A = np.vstack([A[1:, :], b[None, :]])  # methods.jl:231-236
    c = np.concatenate([c[1:], [1.0]])
    return ExplicitRungeKuttaMethod(A, b, c, r)


def RK44():  # RKMethods.jl:515-521
    A = [[0, 0, 0, 0], [0.5, 0, 0, 0], [0, 0.5, 0, 0], [0, 0, 1, 0]]
    b = [1 / 6, 1 / 3, 1 / 3, 1 / 6]
    c = np.sum(np.array(A), axis=1)
    return runge_kutta_method(A, b, c)


def Wray3():  # RKMethods.jl:137-147
    a31 = 8 / 15 - 17 / 60
    A = [[0, 0, 0], [8 / 15, 0, 0], [a31, 5 / 12, 0]]
    b = [a31, 0, 3 / 4]
    c = [0, 8 / 15, a31 + 5 / 12]
    return runge_kutta_method(A, b, c)


def SSP33():  # RKMethods.jl:73-78
    A = [[0, 0, 0], [1, 0, 0], [1 / 4, 1 / 4, 0]]
    return runge_kutta_method(A, [1 / 6, 1 / 6, 2 / 3], np.sum(np.array(A), axis=1))


def FE11():  # RKMethods.jl:45-50
    return runge_kutta_method([[0.0]], [1.0], [0.0])


def ode_method_cache(method, setup):  # time_stepper_caches.jl:34-49
    ns = len(method.b)
    return dict(
        ustart=vectorfield(setup), ku=[vectorfield(setup) for _ in range(ns)], p=scalarfield(setup)
    )


def timestep_(method, stepper, dt, cache):
    """step_explicit_runge_kutta.jl:4-59.  `stepper` is a dict (setup, psolver, u, t, n)."""
    setup, psolver, u, t, n = (stepper[k] for k in ("setup", "psolver", "u", "t", "n"))
    A, b, c = method.A, method.b, method.c
    ustart, ku, p = cache["ustart"], cache["ku"], cache["p"]
    nstage = len(b)
    tstart = t
    ustart[...] = u
    for i in range(nstage):
        apply_bc_u_(u, t, setup)
        momentum_(ku[i], u, None, t, setup)
        t = tstart + c[i] * dt
        u[...] = ustart
        for j in range(i + 1):
            u += dt * A[i, j] * ku[j]
        apply_bc_u_(u, t, setup)
        project_(u, setup, psolver, p)
    apply_bc_u_(u, t, setup)
    return dict(setup=setup, psolver=psolver, u=u, t=t, n=n + 1)


def timestep_lmwray3_(stepper, dt, cache):
    """step_lmwray3.jl:4-107 (low-storage Wray RK3; closure_model = temp = nothing)."""
    setup, psolver, u, n = (stepper[k] for k in ("setup", "psolver", "u", "n"))
    a, b, c = (8 / 15, 5 / 12, 3 / 4), (1 / 4, 0.0), (0.0, 8 / 15, 2 / 3)
    ustart, ku, p = cache["ustart"], cache["ku"][0], cache["p"]
    tstart = stepper["t"]
    ustart[...] = u
    t = tstart
    for i in range(3):
        t = tstart + c[i] * dt
        apply_bc_u_(u, t, setup)
        momentum_(ku, u, None, t, setup)
        u[...] = ustart
        u += a[i] * dt * ku
        apply_bc_u_(u, t, setup)
        project_(u, setup, psolver, p)
        if i != 2:
            ustart += b[i] * dt * ku
    t = tstart + dt
    apply_bc_u_(u, t, setup)
    return dict(setup=setup, psolver=psolver, u=u, t=t, n=n + 1)


def right_hand_side(u, setup, psolver, t):
    """sciml.jl:13-19 / 35-47: project(bc_dudt(momentum(bc(u))))."""
    tmp = apply_bc_u(u, t, setup)
    F = momentum(tmp, None, t, setup)
    apply_bc_u_(F, t, setup, dudt=True)
    project_(F, setup, psolver, scalarfield(setup))
    return F


def get_cfl_timestep(u, setup):  # solver.jl:101-125
    g = setup.grid
    D = g.D
    dt = math.inf
    for a in range(D):
        lo, hi = g.Iu[a][a]
        damin = float(np.min(g.dxu[a][lo:hi]))
        dt_diff = setup.Re * damin**2 / 2
        R = g.Iu[a]
        with np.errstate(divide="ignore"):
            buf = _vec(g.dxu[a], R[a], a, D) / np.abs(u[_sl(R) + (a,)])
        dt = min(dt, dt_diff, float(np.min(buf)))
    return dt


def solve_unsteady(setup, tlims, ustart, method=None, psolver=None, dt=None, cfl=0.9, cache=None):
    """solver.jl:18-92 without processors."""
    method = method or RK44()
    psolver = psolver or default_psolver(setup)
    cache = cache or ode_method_cache(method, setup)
    tstart, tend = tlims
    stepper = dict(setup=setup, psolver=psolver, u=ustart.copy(order="F"), t=tstart, n=0)
    if dt is None:
        while stepper["t"] < tend:
            h = cfl * get_cfl_timestep(stepper["u"], setup)
            h = min(h, tend - stepper["t"])
            stepper = timestep_(method, stepper, h, cache)
    else:
        nstep = int(round((tend - tstart) / dt))
        h = (tend - tstart) / nstep
        for _ in range(nstep):
            stepper = timestep_(method, stepper, h, cache)
    return stepper


# --------------------------------------------------------------------------------------
# Initial conditions                                                   initializers.jl
# --------------------------------------------------------------------------------------
def velocityfield(setup, ufunc, t=0.0, psolver=None, doproject=True):  # initializers.jl:13-46
    g = setup.grid
    D = g.D
    u = vectorfield(setup)
    for a in range(D):
        xs = []
        for b in range(D):
            lo, hi = g.Iu[a][b]
            shape = [1] * D
            shape[b] = hi - lo
            xs.append(g.xu[a][b][lo:hi].reshape(shape))
        u[_sl(g.Iu[a]) + (a,)] = np.broadcast_to(
            ufunc(a, *xs), tuple(hi - lo for lo, hi in g.Iu[a])
        )
    apply_bc_u_(u, t, setup)
    if doproject:
        psolver = psolver or default_psolver(setup)
        u = project(u, setup, psolver)
        apply_bc_u_(u, t, setup)
    return np.asfortranarray(u)


def random_field(setup, t=0.0, A=1.0, kp=10, psolver=None, seed=0):
    """initializers.jl:82-219 with numpy's PCG64 in place of Julia's Xoshiro: same spectrum
    E(k) ∝ k⁴ exp(-2π (k/kp)²) and same construction, *statistically* (not bitwise) equivalent
    (SURVEY.md §8c vi)."""
    assert_uniform_periodic(setup, "Random field")
    g = setup.grid
    D = g.D
    rng = np.random.default_rng(seed)
    tau = 2 * np.pi
    K = tuple((n - 2) // 2 for n in g.N)

    def axis_vec(n, a):
        shape = [1] * D
        shape[a] = n
        return np.arange(n, dtype=np.float64).reshape(shape)

    k = np.zeros(K)
    for a in range(D):
        k = k + axis_vec(K[a], a) ** 2
    k = np.sqrt(k)
    Amag = (8 * tau / 3) / kp**5
    amp = np.sqrt(Amag * k**4 * np.exp(-tau * (k / kp) ** 2)).astype(np.complex128)
    amp *= float(np.prod(g.N))
    xi = [rng.random(K) for _ in range(D)]
    for a in range(D):
        amp = np.concatenate([amp, np.flip(amp, axis=a)], axis=a)
        xi = [
            np.concatenate([xb, np.flip((-1 if a == b else 1) * xb, axis=a)], axis=a)
            for b, xb in enumerate(xi)
        ]
    xis = sum(xi)
    amp = np.exp(1j * tau * xis) * amp
    KK = tuple(2 * kk for kk in K)
    kvec = [np.broadcast_to(axis_vec(KK[a], a), KK) for a in range(D)]
    knorm = np.sqrt(sum(kv**2 for kv in kvec))
    if D == 2:
        th = rng.random(KK)
        e = [np.cos(tau * th), np.sin(tau * th)]
    else:
        th = rng.random(KK)
        ph = rng.random(KK)
        e = [
            np.sin(np.pi * th) * np.cos(tau * ph),
            np.sin(np.pi * th) * np.sin(tau * ph),
            np.cos(np.pi * th),
        ]
    ke = sum(e[a] * kvec[a] for a in range(D))
    for a in range(D):
        e0 = e[a][(0,) * D]
        with np.errstate(divide="ignore", invalid="ignore"):
            e[a] = e[a] - kvec[a] * ke / knorm**2
        e[a][(0,) * D] = e0
    enorm = np.sqrt(sum(ea**2 for ea in e))
    e = [ea / enorm for ea in e]
    uhat = np.stack([amp * ea for ea in e], axis=-1)
    uin = A * np.real(np.fft.ifftn(uhat, axes=tuple(range(D))))
    u = vectorfield(setup)
    u[...] = np.pad(uin, [(1, 1)] * D + [(0, 0)], mode="wrap")
    apply_bc_u_(u, t, setup)
    psolver = psolver or default_psolver(setup)
    u = project(u, setup, psolver)
    apply_bc_u_(u, t, setup)
    return np.asfortranarray(u)


# ------------------------------------------------------------------ canned problem setups
def tgv2d_ufunc(Re):
    """examples/TaylorGreenVortex2D.jl:62-63"""

    def sol(t):
        def f(a, x, y):
            return (-np.sin(x) * np.cos(y) if a == 0 else np.cos(x) * np.sin(y)) * math.exp(
                -2 * t / Re
            )

        return f

    return sol


def tgv3d_ufunc(a, x, y, z):
    """examples/TaylorGreenVortex3D.jl:30-37"""
    if a == 0:
        return np.sin(tau_ * x) * np.cos(tau_ * y) * np.sin(tau_ * z) / 2
    if a == 1:
        return -np.cos(tau_ * x) * np.sin(tau_ * y) * np.sin(tau_ * z) / 2
    return np.zeros(np.broadcast(x, y, z).shape)


tau_ = 2 * np.pi


# ======================================================================================
# field diagnostics, temperature equation, body force, Smagorinsky closure.
# The reference's tests hold no known-answer values for these operators (test/operators.jl checks them through
# ChainRules only), so beyond the analytic checks in tests/test_oracle_fields.py their parity is "unpinned": the
# restatement below follows the cited lines term by term.
# ======================================================================================
def _full_ranges(N, trim=0):
    return tuple((0, n - trim) for n in N)


def vorticity_(w, u, setup):  # operators.jl:985-1020   (ndrange = N .- 1)
    g = setup.grid
    D = g.D
    R = _full_ranges(g.N, 1)

    def d(comp, axis):  # (u[I+e(axis), comp] - u[I, comp]) / Δu[axis][I[axis]]
        return (u[_sl(R, _e(D, axis)) + (comp,)] - u[_sl(R) + (comp,)]) / _vec(g.dxu[axis], R[axis], axis, D)

    if D == 2:
        w[_sl(R)] = d(1, 0) - d(0, 1)
    else:
        for a, ap, am in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
            w[_sl(R) + (a,)] = d(am, ap) - d(ap, am)
    return w


def vorticity(u, setup):
    return vorticity_(scalarfield(setup) if setup.grid.D == 2 else vectorfield(setup), u, setup)


def interpolate_u_p_(up, u, setup):  # operators.jl:1311-1326
    g = setup.grid
    D, R = g.D, g.Ip
    for a in range(D):
        up[_sl(R) + (a,)] = (u[_sl(R, _e(D, a, -1)) + (a,)] + u[_sl(R) + (a,)]) / 2
    return up


def interpolate_u_p(u, setup):
    return interpolate_u_p_(vectorfield(setup), u, setup)


def interpolate_w_p_(wp, w, setup):  # operators.jl:1336-1370  (interpolate_ω_p!)
    g = setup.grid
    D, R = g.D, g.Ip
    if D == 2:
        wp[_sl(R)] = (w[_sl(R, (-1, -1))] + w[_sl(R)]) / 2
    else:
        for a in range(3):
            ap, am = (a + 1) % 3, (a + 2) % 3
            sh = _add(_e(D, ap, -1), _e(D, am, -1))
            wp[_sl(R) + (a,)] = (w[_sl(R, sh) + (a,)] + w[_sl(R) + (a,)]) / 2
    return wp


def interpolate_w_p(w, setup):
    return interpolate_w_p_(scalarfield(setup) if setup.grid.D == 2 else vectorfield(setup), w, setup)


def Dfield_(d, G, p, setup, eps=EPS):  # operators.jl:1385-1422
    g = setup.grid
    D, R = g.D, g.Ip
    pressuregradient_(G, p, setup)
    gg = 0.0
    lap = 0.0
    for a in range(D):
        Gm, Gc = G[_sl(R, _e(D, a, -1)) + (a,)], G[_sl(R) + (a,)]
        gg = gg + (Gm + Gc) ** 2
        lap = lap + (Gc - Gm) / _vec(g.dx[a], R[a], a, D)
    lap = np.where(lap > 0, np.maximum(lap, eps), np.minimum(lap, -eps))
    d[_sl(R)] = np.sqrt(gg) / 2 / lap
    return d


def Dfield(p, setup, eps=EPS):
    return Dfield_(scalarfield(setup), vectorfield(setup), p, setup, eps)


def Qfield_(Q, u, setup):  # operators.jl:1440-1460
    g = setup.grid
    D, R = g.D, g.Ip
    q = 0.0
    for a in range(D):
        for b in range(D):
            dab = (u[_sl(R) + (a,)] - u[_sl(R, _e(D, b, -1)) + (a,)]) / _vec(g.dx[b], R[b], b, D)
            dba = (u[_sl(R) + (b,)] - u[_sl(R, _e(D, a, -1)) + (b,)]) / _vec(g.dx[a], R[a], a, D)
            q = q - dab * dba / 2
    Q[_sl(R)] = q
    return Q


def Qfield(u, setup):
    return Qfield_(scalarfield(setup), u, setup)


def gradu(u, setup):
    """∇(u, I, Δ, Δu) at every pressure point (operators.jl:1023-1034, 1069-1085): array Ip-shape + (D, D), [..., a, b] = ∂u^a/∂x^b."""
    g = setup.grid
    D, R = g.D, g.Ip
    out = np.zeros(tuple(hi - lo for lo, hi in R) + (D, D))
    for a in range(D):
        for b in range(D):
            if a == b:
                v = (u[_sl(R) + (a,)] - u[_sl(R, _e(D, b, -1)) + (a,)]) / _vec(g.dx[b], R[b], b, D)
            else:
                ea, eb = _e(D, a, -1), _e(D, b)
                emb = _e(D, b, -1)
                hb = _vec(g.dxu[b], R[b], b, D)
                hbm = _vec(g.dxu[b], R[b], b, D, -1)
                v = (
                    (u[_sl(R, eb) + (a,)] - u[_sl(R) + (a,)]) / hb
                    + (u[_sl(R, _add(ea, eb)) + (a,)] - u[_sl(R, ea) + (a,)]) / hb
                    + (u[_sl(R) + (a,)] - u[_sl(R, emb) + (a,)]) / hbm
                    + (u[_sl(R, ea) + (a,)] - u[_sl(R, _add(ea, emb)) + (a,)]) / hbm
                ) / 4
            out[..., a, b] = v
    return out


def dissipation_from_strain_(eps_out, u, setup):  # operators.jl:836-854
    g = setup.grid
    G = gradu(u, setup)
    S = (G + np.swapaxes(G, -1, -2)) / 2
    eps_out[_sl(g.Ip)] = 2 * (1.0 / setup.Re) * np.sum(S * S, axis=(-1, -2))
    return eps_out


def dissipation_from_strain(u, setup):
    return dissipation_from_strain_(scalarfield(setup), u, setup)


def eig2field_(lam, u, setup):  # operators.jl:1472-1492   (3-D only)
    g = setup.grid
    assert g.D == 3
    G = gradu(u, setup)
    S = (G + np.swapaxes(G, -1, -2)) / 2
    Rm = (G - np.swapaxes(G, -1, -2)) / 2
    M = S @ S + Rm @ Rm
    lam[_sl(g.Ip)] = np.linalg.eigvalsh(M)[..., 1]
    return lam


def eig2field(u, setup):
    return eig2field_(scalarfield(setup), u, setup)


# ---- temperature equation --------------------------------------------------------------------------------------------
@dataclass
class Temperature:  # setup.jl:48-87 (gdir 0-based here)
    a1: float
    a2: float
    a3: float
    a4: float
    gamma: float
    dodissipation: bool
    boundary_conditions: tuple
    gdir: int


def temperature_equation(Pr, Ra, Ge, boundary_conditions, dodissipation=True, gdir=1, nondim_type=1):
    if nondim_type == 1:
        a1, a2, a3, a4 = math.sqrt(Pr / Ra), 1.0, Ge * math.sqrt(Pr / Ra), 1 / math.sqrt(Pr * Ra)
    elif nondim_type == 2:
        a1, a2, a3, a4 = Pr, Pr * Ra, Ge / Ra, 1.0
    elif nondim_type == 3:
        a1, a2, a3, a4 = math.sqrt(Pr * Ge / Ra), Ge, math.sqrt(Pr * Ge / Ra), math.sqrt(Ge / (Pr * Ra))
    else:
        raise ValueError(nondim_type)
    return Temperature(a1, a2, a3, a4, a1 / a3, bool(dodissipation), tuple(boundary_conditions), int(gdir))


def apply_bc_temp_(temp, t, setup):  # boundary_conditions.jl:236-246, 338-339, 391-405, 466-467, 512-513
    g = setup.grid
    D, N = g.D, g.N
    bcs = setup.temperature.boundary_conditions
    for be in range(D):
        for isright in (False, True):
            bc = bcs[be][int(isright)]
            lo, hi = g.Ip[be]
            i = hi if isright else lo - 1
            if isinstance(bc, PeriodicBC):
                if isright:
                    continue
                temp[_plane(N, be, lo - 1)] = temp[_plane(N, be, hi - 1)]
                temp[_plane(N, be, hi)] = temp[_plane(N, be, lo)]
            elif isinstance(bc, DirichletBC):
                # `boundary(β, N, Ip, isright)` (boundary_conditions.jl:97-103) spans 1:N in the other directions
                if bc.u is None:
                    val = 0.0
                elif callable(bc.u):
                    xs = []
                    for b in range(D):
                        shape = [1] * D
                        if b == be:
                            xs.append(g.xp[b][i : i + 1].reshape(shape))
                        else:
                            shape[b] = N[b]
                            xs.append(g.xp[b].reshape(shape))
                    val = bc.u(*xs, t)
                else:
                    val = float(bc.u)
                temp[_plane(N, be, i)] = val
            elif isinstance(bc, (SymmetricBC, PressureBC)):
                j = i - 1 if isright else i + 1
                temp[_plane(N, be, i)] = temp[_plane(N, be, j)]
            else:
                raise TypeError(bc)
    return temp


def apply_bc_temp(temp, t, setup):
    return apply_bc_temp_(temp.copy(order="F"), t, setup)


def temperaturefield(setup, tempfunc, t=0.0):  # initializers.jl:49-57
    g = setup.grid
    D = g.D
    xs = []
    for b in range(D):
        shape = [1] * D
        shape[b] = g.Ip[b][1] - g.Ip[b][0]
        xs.append(g.xp[b][g.Ip[b][0] : g.Ip[b][1]].reshape(shape))
    temp = scalarfield(setup)
    temp[_sl(g.Ip)] = np.broadcast_to(tempfunc(*xs), tuple(hi - lo for lo, hi in g.Ip))
    return apply_bc_temp_(temp, t, setup)


def _avg(phi, setup, R, a, shift=None):  # operators.jl:59-62 at I + shift, I over R
    g = setup.grid
    D = g.D
    shift = shift or (0,) * D
    s = shift[a]
    d0 = _vec(g.dx[a], R[a], a, D, s)
    d1 = _vec(g.dx[a], R[a], a, D, s + 1)
    return (d1 * phi[_sl(R, shift)] + d0 * phi[_sl(R, _add(shift, _e(D, a)))]) / (d0 + d1)


def convection_diffusion_temp_(c, u, temp, setup):  # operators.jl:712-737   (adds to c)
    g = setup.grid
    D, R = g.D, g.Ip
    a4 = setup.temperature.a4
    acc = 0.0
    for b in range(D):
        em = _e(D, b, -1)
        dT1 = (temp[_sl(R)] - temp[_sl(R, em)]) / _vec(g.dxu[b], R[b], b, D, -1)
        dT2 = (temp[_sl(R, _e(D, b))] - temp[_sl(R)]) / _vec(g.dxu[b], R[b], b, D)
        uT1 = u[_sl(R, em) + (b,)] * _avg(temp, setup, R, b, em)
        uT2 = u[_sl(R) + (b,)] * _avg(temp, setup, R, b)
        acc = acc + (-(uT2 - uT1) + a4 * (dT2 - dT1)) / _vec(g.dx[b], R[b], b, D)
    c[_sl(R)] += acc
    return c


def convection_diffusion_temp(u, temp, setup):
    return convection_diffusion_temp_(scalarfield(setup), u, temp, setup)


def dissipation_(diss, diff, u, setup):  # operators.jl:791-814   (adds to diss; diff is scratch)
    g = setup.grid
    D, R = g.D, g.Ip
    T = setup.temperature
    diff[...] = 0.0
    diffusion_(diff, u, setup)
    acc = 0.0
    for b in range(D):
        em = _e(D, b, -1)
        acc = acc + setup.Re * T.a1 / T.gamma * (u[_sl(R, em) + (b,)] * diff[_sl(R, em) + (b,)] + u[_sl(R) + (b,)] * diff[_sl(R) + (b,)]) / 2
    diss[_sl(R)] += acc
    return diss


def dissipation(u, setup):
    return dissipation_(scalarfield(setup), vectorfield(setup), u, setup)


def gravity_(F, temp, setup):  # operators.jl:914-931   (adds to F; the whole Iu[gdir] range)
    g = setup.grid
    T = setup.temperature
    R = g.Iu[T.gdir]
    F[_sl(R) + (T.gdir,)] += T.a2 * _avg(temp, setup, R, T.gdir)
    return F


def gravity(temp, setup):
    return gravity_(vectorfield(setup), temp, setup)


# ---- body force -------------------------------------------------------------------------------------------------------
def bodyforce_field(setup, f, t):  # operators.jl:878-897: f(α, x..., t) on the FULL padded coordinate vectors xu[α]
    g = setup.grid
    D = g.D
    F = vectorfield(setup)
    for a in range(D):
        xs = []
        for b in range(D):
            shape = [1] * D
            shape[b] = g.N[b]
            xs.append(g.xu[a][b].reshape(shape))
        F[..., a] = np.broadcast_to(f(a, *xs, t), g.N)
    return F


def applybodyforce_(F, u, t, setup):
    if setup.issteadybodyforce:
        F += setup.bodyforce
    else:
        F += bodyforce_field(setup, setup.bodyforce, t)
    return F


def make_setup_ext(x, boundary_conditions=None, Re=None, bodyforce=None, issteadybodyforce=True, closure_model=None, temperature=None):
    """Setup(...) with the optional pieces (setup.jl:2-46): Re defaults to 1/α1 with a temperature equation; a steady body force is
    evaluated once at t = 0."""
    if Re is None:
        Re = 1000.0 if temperature is None else 1.0 / temperature.a1
    s = make_setup(x, boundary_conditions, Re)
    s.temperature = temperature
    s.closure_model = closure_model
    s.bodyforce = bodyforce
    s.issteadybodyforce = False
    if bodyforce is not None and issteadybodyforce:
        s.bodyforce = bodyforce_field(s, bodyforce, 0.0)
        s.issteadybodyforce = True
    return s


def momentum_ext_(F, u, temp, t, setup):  # operators.jl:967-976
    F[...] = 0.0
    convectiondiffusion_(F, u, setup)
    if setup.bodyforce is not None:
        applybodyforce_(F, u, t, setup)
    if temp is not None:
        gravity_(F, temp, setup)
    return F


# ---- Smagorinsky closure ----------------------------------------------------------------------------------------------
def smagtensor_(sig, u, theta, setup):  # operators.jl:1135-1150; sig: N + (D, D)
    g = setup.grid
    D, R = g.D, g.Ip
    G = gradu(u, setup)
    S = (G + np.swapaxes(G, -1, -2)) / 2
    d2 = 0.0
    for a in range(D):
        d2 = d2 + _vec(g.dx[a], R[a], a, D) ** 2
    d = np.sqrt(d2)
    eddy = theta**2 * d**2 * np.sqrt(2 * np.sum(S * S, axis=(-1, -2)))
    sig[_sl(R)] = 2 * eddy[..., None, None] * S
    return sig


def divoftensor_(s, sig, setup):  # operators.jl:1203-1236
    g = setup.grid
    D = g.D
    for a in range(D):
        R = g.Iu[a]
        acc = 0.0
        for b in range(D):
            h = _vec(g.dxu[b] if a == b else g.dx[b], R[b], b, D)
            ea, eb, emb = _e(D, a), _e(D, b), _e(D, b, -1)
            if a == b:
                s2 = sig[_sl(R, eb) + (a, b)]
                s1 = sig[_sl(R) + (a, b)]
            else:
                s2 = (sig[_sl(R) + (a, b)] + sig[_sl(R, eb) + (a, b)] + sig[_sl(R, _add(ea, eb)) + (a, b)] + sig[_sl(R, ea) + (a, b)]) / 4
                s1 = (sig[_sl(R, emb) + (a, b)] + sig[_sl(R) + (a, b)] + sig[_sl(R, _add(ea, emb)) + (a, b)] + sig[_sl(R, ea) + (a, b)]) / 4
            acc = acc + (s2 - s1) / h
        s[_sl(R) + (a,)] = acc
    return s


def smagorinsky_closure(setup):  # operators.jl:1289-1300
    g = setup.grid
    D = g.D
    sig = np.zeros(g.N + (D, D), order="F")
    s = vectorfield(setup)

    def closure(u, theta):
        smagtensor_(sig, u, theta, setup)
        for i in range(D):
            for j in range(D):
                comp = np.asfortranarray(sig[..., i, j])
                apply_bc_p_(comp, 0.0, setup)
                sig[..., i, j] = comp
        return divoftensor_(s, sig, setup)

    return closure


# ---- steppers with temperature / closure / body force -----------------------------------------------------------------
def ode_method_cache_ext(method, setup):  # time_stepper_caches.jl:34-49
    ns = len(method.b)
    c = ode_method_cache(method, setup)
    if setup.temperature is not None:
        c.update(tempstart=scalarfield(setup), ktemp=[scalarfield(setup) for _ in range(ns)], diff=vectorfield(setup))
    return c


def timestep_ext_(method, stepper, dt, cache, theta=None):
    """step_explicit_runge_kutta.jl:4-59 with temperature, closure model and body force."""
    setup, psolver, u, temp, t, n = (stepper[k] for k in ("setup", "psolver", "u", "temp", "t", "n"))
    A, b, c = method.A, method.b, method.c
    ustart, ku, p = cache["ustart"], cache["ku"], cache["p"]
    m = setup.closure_model
    T = setup.temperature
    nstage = len(b)
    tstart = t
    ustart[...] = u
    if temp is not None:
        cache["tempstart"][...] = temp
    for i in range(nstage):
        apply_bc_u_(u, t, setup)
        if temp is not None:
            apply_bc_temp_(temp, t, setup)
        momentum_ext_(ku[i], u, temp, t, setup)
        if temp is not None:
            kt = cache["ktemp"][i]
            kt[...] = 0.0
            convection_diffusion_temp_(kt, u, temp, setup)
            if T.dodissipation:
                dissipation_(kt, cache["diff"], u, setup)
        if m is not None:
            ku[i] += m(u, theta)
        t = tstart + c[i] * dt
        u[...] = ustart
        for j in range(i + 1):
            u += dt * A[i, j] * ku[j]
        if temp is not None:
            temp[...] = cache["tempstart"]
            for j in range(i + 1):
                temp += dt * A[i, j] * cache["ktemp"][j]
        apply_bc_u_(u, t, setup)
        project_(u, setup, psolver, p)
    apply_bc_u_(u, t, setup)
    if temp is not None:
        apply_bc_temp_(temp, t, setup)
    return dict(setup=setup, psolver=psolver, u=u, temp=temp, t=t, n=n + 1)


def timestep_lmwray3_ext_(stepper, dt, cache, theta=None):
    """step_lmwray3.jl:4-107 with temperature, closure model and body force."""
    setup, psolver, u, temp, n = (stepper[k] for k in ("setup", "psolver", "u", "temp", "n"))
    a, b, c = (8 / 15, 5 / 12, 3 / 4), (1 / 4, 0.0), (0.0, 8 / 15, 2 / 3)
    ustart, ku, p = cache["ustart"], cache["ku"][0], cache["p"]
    m = setup.closure_model
    T = setup.temperature
    tstart = stepper["t"]
    ustart[...] = u
    if temp is not None:
        tempstart, ktemp = cache["tempstart"], cache["ktemp"][0]
        tempstart[...] = temp
    t = tstart
    for i in range(3):
        t = tstart + c[i] * dt
        apply_bc_u_(u, t, setup)
        if temp is not None:
            apply_bc_temp_(temp, t, setup)
        momentum_ext_(ku, u, temp, t, setup)
        if m is not None:
            ku += m(u, theta)
        if temp is not None:
            ktemp[...] = 0.0
            convection_diffusion_temp_(ktemp, u, temp, setup)
            if T.dodissipation:
                dissipation_(ktemp, cache["diff"], u, setup)
        u[...] = ustart
        u += a[i] * dt * ku
        if temp is not None:
            temp[...] = tempstart
            temp += a[i] * dt * ktemp
        apply_bc_u_(u, t, setup)
        project_(u, setup, psolver, p)
        if i != 2:
            ustart += b[i] * dt * ku
            if temp is not None:
                tempstart += b[i] * dt * ktemp
    t = tstart + dt
    apply_bc_u_(u, t, setup)
    if temp is not None:
        apply_bc_temp_(temp, t, setup)
    return dict(setup=setup, psolver=psolver, u=u, temp=temp, t=t, n=n + 1)


# ---- energy spectrum ---------------------------------------------------------------------------------------------------
def spectral_stuff(setup, npoint=100, a=(1 + math.sqrt(5)) / 2):  # utils.jl:49-108
    g = setup.grid
    D = g.D
    K = tuple(n // 2 for n in g.Np)
    ks = np.meshgrid(*[np.arange(k, dtype=np.float64) for k in K], indexing="ij")
    k = np.sqrt(sum(x**2 for x in ks)).reshape(-1, order="F")
    kmax = min(K) - 1
    isort = np.argsort(k, kind="stable")
    ksort = k[isort]
    kap = np.exp(np.linspace(0.0, math.log(kmax), npoint))  # logrange(1, kmax, npoint)
    kap = np.unique(np.rint(kap).astype(np.int64))
    inds = []
    for ki in kap:
        if D == 2:  # dyadic binning
            lo, hi = ki / a, ki * a
        else:  # linear binning
            lo, hi = ki - 0.01, ki + 1 - 0.01
        jstart = int(np.searchsorted(ksort, lo, side="left"))
        jstop = int(np.searchsorted(ksort, hi, side="left"))
        inds.append(isort[jstart:jstop])
    return inds, kap, K


def observespectrum(u, setup, npoint=100, a=(1 + math.sqrt(5)) / 2):  # processors.jl:303-332
    g = setup.grid
    D = g.D
    inds, kap, K = spectral_stuff(setup, npoint, a)
    e = 0.0
    for c in range(D):
        uhat = np.fft.fftn(u[_sl(g.Ip) + (c,)])
        half = uhat[tuple(slice(0, k) for k in K)]
        e = e + np.abs(half) ** 2 / (2 * float(np.prod(g.Np)) ** 2)
    e = e.reshape(-1, order="F")
    return np.array([e[i].sum() for i in inds]), kap


def get_scale_numbers(u, setup):  # operators.jl:1558-1617, as written (its uavg sums all components under each component's weights)
    g = setup.grid
    D = g.D
    visc = 1.0 / setup.Re

    def outer(vs):
        w = 1.0
        for b, v in enumerate(vs):
            shape = [1] * D
            shape[b] = len(v)
            w = w * np.asarray(v).reshape(shape)
        return np.broadcast_to(w, g.N)

    Om = outer([g.dx[b] for b in range(D)])
    uavg2 = 0.0
    for a in range(D):
        Omu = outer([g.dxu[b] if a == b else g.dx[b] for b in range(D)])
        field = u**2 * Omu[..., None]
        uavg2 += field[_sl(g.Iu[0])].sum() / Omu[_sl(g.Iu[0])].sum()
    uavg = math.sqrt(uavg2)
    eps = dissipation_from_strain(u, setup)
    eps = (Om * eps)[_sl(g.Ip)].sum() / Om[_sl(g.Ip)].sum()
    eta = (visc**3 / eps) ** 0.25
    lam = math.sqrt(5 * visc / eps) * uavg
    Relam = lam * uavg / math.sqrt(3.0) / visc
    assert_uniform_periodic(setup, "Scale numbers")
    K = tuple(n // 2 for n in g.Np)
    up = u[_sl(g.Ip)]
    uhat = np.fft.fftn(up, axes=tuple(range(D)))[tuple(slice(0, k) for k in K)]
    e = np.abs(uhat) ** 2 / (2 * float(np.prod(g.Np)) ** 2)
    ks = np.meshgrid(*[np.arange(k, dtype=np.float64) for k in K], indexing="ij")
    kk = np.sqrt(sum(x**2 for x in ks))
    with np.errstate(divide="ignore", invalid="ignore"):
        e = e / kk[..., None]
    e = e.sum(axis=-1)
    e[(0,) * D] = 0.0
    L = 3 * math.pi / 2 / uavg**2 * e.sum()
    return dict(uavg=uavg, eps=eps, eta=eta, lam=lam, Relam=Relam, L=L, tau=L / uavg, Re_int=L * uavg / visc)


def tensorbasis(u, setup):  # tensorbasis.jl:16-72; returns B: N + (nb, D, D), V: N + (nv,)
    g = setup.grid
    D = g.D
    G = gradu(u, setup)
    S = (G + np.swapaxes(G, -1, -2)) / 2
    R = (G - np.swapaxes(G, -1, -2)) / 2
    Id = np.broadcast_to(np.eye(D), S.shape)
    tr = lambda X: np.trace(X, axis1=-2, axis2=-1)
    if D == 2:
        Bs = [Id, S, S @ R - R @ S]
        Vs = [np.sum(S * S, axis=(-1, -2)), np.sum(R * R, axis=(-1, -2))]
    else:
        Bs = [Id, S, S @ R - R @ S, S @ S, R @ R, S @ S @ R - R @ S @ S, S @ R @ R + R @ R @ S, R @ S @ R @ R - R @ R @ S @ R,
              S @ R @ S @ S - S @ S @ R @ S, S @ S @ R @ R + R @ R @ S @ S, R @ S @ S @ R @ R - R @ R @ S @ S @ R]
        Vs = [tr(S @ S), tr(R @ R), tr(S @ S @ S), tr(S @ R @ R), tr(S @ S @ R @ R)]
    B = np.zeros(g.N + (len(Bs), D, D))
    V = np.zeros(g.N + (len(Vs),))
    for i, b in enumerate(Bs):
        B[_sl(g.Ip) + (i,)] = b
    for i, v in enumerate(Vs):
        V[_sl(g.Ip) + (i,)] = v
    return B, V
