"""
Sparse-matrix twins of the linear operators — the reference's SECOND, independent implementation (src/matrices.jl), restated.

*** TEST INFRASTRUCTURE ONLY *** (same rule as oracle/ins_oracle.py: tests/, smoke() and bench.py's cpu_baseline leg only).

The reference tests its stencil kernels against matrices assembled by index arithmetic (test/matrices.jl:19-51: BC matrices, divergence, pressure
gradient, diffusion; test/operators.jl:90-105: `laplacian!` against `laplacian_mat`), and factorises `laplacian_mat` in `psolver_direct`
(pressure.jl:117-154).  Every function below builds its (i, j, v) triplets from index ranges exactly as the cited lines do and NEVER calls the stencil
code of ins_oracle.py (it reads the grid's index ranges and metric vectors only), so "operator == matrix" compares two implementations, as it does in the
reference.  0-based indices; column-major (Fortran) linear indices like Julia's `reshape(1:n, N...)`.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from . import ins_oracle as o

EPS = o.EPS


def _ilin(shape):
    return np.arange(int(np.prod(shape))).reshape(shape, order="F")


def _take(ilin, ranges, tail=()):
    """ilin[ranges..., tail...][:] in column-major order; a range is a (lo, hi) pair, an index array or None (= all); tail entries are ints / arrays."""
    axes = []
    for d, r in enumerate(list(ranges) + list(tail)):
        if r is None:
            axes.append(np.arange(ilin.shape[d]))
        elif isinstance(r, tuple):
            axes.append(np.arange(r[0], r[1]))
        else:
            axes.append(np.atleast_1d(np.asarray(r, dtype=int)))
    return ilin[np.ix_(*axes)].reshape(-1, order="F")


def _boundary(be, N, I, isright):
    """boundary_conditions.jl:97-103: the plane just outside the index range I in direction β (whole extent in the other directions)."""
    lo, hi = I[be]
    i = hi if isright else lo - 1
    return [(i, i + 1) if d == be else (0, N[d]) for d in range(len(N))]


def _shift(ranges, be, s):
    return [(lo + s, hi + s) if d == be else (lo, hi) for d, (lo, hi) in enumerate(ranges)]


def _coo(i, j, v, shape):
    i, j = np.concatenate(i), np.concatenate(j)
    v = np.ones(len(i)) if v is None else np.concatenate(v)
    return sp.coo_matrix((v, (i, j)), shape=shape).tocsr()


def _without(I_be, N_be, isright):
    """matrices.jl:186-190: every index of the direction except the boundary one (unused ghost volumes included)."""
    lo, hi = I_be
    drop = hi if isright else lo - 1
    return np.array([k for k in range(N_be) if k != drop], dtype=int)


# ------------------------------------------------------------------------------------------------ padding   matrices.jl:23-53
def pad_scalarfield_mat(setup):
    g = setup.grid
    n, npp = int(np.prod(g.N)), int(np.prod(g.Np))
    i = _take(_ilin(g.N), list(g.Ip))
    return sp.coo_matrix((np.ones(npp), (i, np.arange(npp))), shape=(n, npp)).tocsr()


def pad_vectorfield_mat(setup):
    g = setup.grid
    D = g.D
    n = int(np.prod(g.N)) * D
    ilin = _ilin(g.N + (D,))
    i = np.concatenate([_take(ilin, list(g.Iu[a]), (a,)) for a in range(D)])
    nu = len(i)
    return sp.coo_matrix((np.ones(nu), (i, np.arange(nu))), shape=(n, nu)).tocsr()


# ------------------------------------------------------------------------------------------------ BC matrices   matrices.jl:68-377
def _bc_u_side(bc, setup, be, isright):
    g = setup.grid
    D, N = g.D, g.N
    n = int(np.prod(N)) * D
    ilin = _ilin(N + (D,))
    comps = np.arange(D)
    i, j = [], []
    if isinstance(bc, o.PeriodicBC):  # matrices.jl:108-137
        if isright:
            return sp.identity(n, format="csr")
        nb = [(1, N[d] - 1) if d == be else None for d in range(D)]
        i.append(_take(ilin, nb, (comps,)))
        j.append(_take(ilin, nb, (comps,)))
        Ia, Ib = _boundary(be, N, g.Ip, False), _boundary(be, N, g.Ip, True)
        Ja, Jb = _shift(Ia, be, +1), _shift(Ib, be, -1)
        i += [_take(ilin, Ia, (comps,)), _take(ilin, Ib, (comps,))]
        j += [_take(ilin, Jb, (comps,)), _take(ilin, Ja, (comps,))]
        return _coo(i, j, None, (n, n))
    for a in range(D):  # Dirichlet matrices.jl:171-201, Symmetric :232-268, Pressure :299-334
        inds = _without(g.Iu[a][be], N[be], isright)
        nb = [inds if d == be else None for d in range(D)]
        i.append(_take(ilin, nb, (a,)))
        j.append(_take(ilin, nb, (a,)))
        copies = isinstance(bc, o.PressureBC) or (isinstance(bc, o.SymmetricBC) and a != be)
        if copies:  # u[I, α] = u[J, α]
            I = _boundary(be, N, g.Iu[a], isright)
            J = _shift(I, be, -1 if isright else +1)
            i.append(_take(ilin, I, (a,)))
            j.append(_take(ilin, J, (a,)))
    return _coo(i, j, None, (n, n))


def _bc_p_side(bc, setup, be, isright):
    g = setup.grid
    D, N = g.D, g.N
    n = int(np.prod(N))
    ilin = _ilin(N)
    i, j = [], []
    if isinstance(bc, o.PeriodicBC):  # matrices.jl:139-167
        if isright:
            return sp.identity(n, format="csr")
        nb = [(1, N[d] - 1) if d == be else None for d in range(D)]
        i.append(_take(ilin, nb))
        j.append(_take(ilin, nb))
        Ia, Ib = _boundary(be, N, g.Ip, False), _boundary(be, N, g.Ip, True)
        Ja, Jb = _shift(Ia, be, +1), _shift(Ib, be, -1)
        i += [_take(ilin, Ia), _take(ilin, Ib)]
        j += [_take(ilin, Jb), _take(ilin, Ja)]
        return _coo(i, j, None, (n, n))
    if isinstance(bc, o.DirichletBC):  # matrices.jl:203
        return sp.identity(n, format="csr")
    inds = _without(g.Ip[be], N[be], isright)  # Symmetric matrices.jl:270-295, Pressure :336-366
    nb = [inds if d == be else None for d in range(D)]
    i.append(_take(ilin, nb))
    j.append(_take(ilin, nb))
    if isinstance(bc, o.SymmetricBC):  # p[I] = p[J]
        I = _boundary(be, N, g.Ip, isright)
        J = _shift(I, be, -1 if isright else +1)
        i.append(_take(ilin, I))
        j.append(_take(ilin, J))
    return _coo(i, j, None, (n, n))


def _compose(side, setup, n):
    B = sp.identity(n, format="csr")
    for be in range(setup.grid.D):  # matrices.jl:72-80: B = b * a * B, direction after direction
        bc_a, bc_b = setup.boundary_conditions[be]
        B = side(bc_b, setup, be, True) @ (side(bc_a, setup, be, False) @ B)
    return B.tocsr()


def bc_u_mat(setup):
    """matrices.jl:68-80.  Only the part that depends on u itself (non-zero Dirichlet data is not part of the matrix)."""
    return _compose(_bc_u_side, setup, int(np.prod(setup.grid.N)) * setup.grid.D)


def bc_p_mat(setup):
    """matrices.jl:82-93."""
    return _compose(_bc_p_side, setup, int(np.prod(setup.grid.N)))


# ------------------------------------------------------------------------------------------------ operators   matrices.jl:379-555
def _metric_on(vec, rng, be, D):
    """map(I -> vec[I[β]], I) over the index box rng, flattened column-major."""
    shape = [hi - lo for lo, hi in rng]
    v = np.asarray(vec)[rng[be][0] : rng[be][1]]
    return np.broadcast_to(v.reshape([-1 if d == be else 1 for d in range(D)]), shape).reshape(-1, order="F")


def divergence_mat(setup):
    """matrices.jl:389-427: div[I] += (u[I, α] − u[I − e(α), α]) / Δ[α][I[α]] over Ip."""
    g = setup.grid
    D, N = g.D, g.N
    n = int(np.prod(N))
    ilp, ilu = _ilin(N), _ilin(N + (D,))
    I = list(g.Ip)
    i, j, v = [], [], []
    for a in range(D):
        dI = _metric_on(g.dx[a], I, a, D)
        i += [_take(ilp, I), _take(ilp, I)]
        j += [_take(ilu, I, (a,)), _take(ilu, _shift(I, a, -1), (a,))]
        v += [1.0 / dI, -1.0 / dI]
    return _coo(i, j, v, (n, n * D))


def pressuregradient_mat(setup):
    """matrices.jl:430-467: G[I, α] = (p[I + e(α)] − p[I]) / Δu[α][I[α]] over Iu[α]."""
    g = setup.grid
    D, N = g.D, g.N
    n = int(np.prod(N))
    ilp, ilu = _ilin(N), _ilin(N + (D,))
    i, j, v = [], [], []
    for a in range(D):
        I = list(g.Iu[a])
        dI = _metric_on(g.dxu[a], I, a, D)
        i += [_take(ilu, I, (a,)), _take(ilu, I, (a,))]
        j += [_take(ilp, _shift(I, a, +1)), _take(ilp, I)]
        v += [1.0 / dI, -1.0 / dI]
    return _coo(i, j, v, (n * D, n))


def volume_mat(setup):
    """matrices.jl:470-477: diag of scalewithvolume!(ones) — the product of the widths Δ over the WHOLE padded array (operators.jl:81-95)."""
    g = setup.grid
    vol = np.ones(g.N)
    for a in range(g.D):
        vol = vol * np.asarray(g.dx[a]).reshape([-1 if d == a else 1 for d in range(g.D)])
    return sp.diags(vol.reshape(-1, order="F"), format="csr")


def laplacian_mat(setup):
    """matrices.jl:483-492: P' Ω M Bu G Bp P on the pressure degrees of freedom."""
    P = pad_scalarfield_mat(setup)
    return (P.T @ volume_mat(setup) @ divergence_mat(setup) @ bc_u_mat(setup) @ pressuregradient_mat(setup) @ bc_p_mat(setup) @ P).tocsr()


def diffusion_mat(setup):
    """matrices.jl:494-555 (no viscosity factor): F[I, α] += (∂b − ∂a) / Δuαβ with the strong zero for widths ≤ 2 eps."""
    g = setup.grid
    D, N = g.D, g.N
    n = int(np.prod(N)) * D
    ilin = _ilin(N + (D,))
    i, j, v = [], [], []
    for a in range(D):
        I = list(g.Iu[a])
        for be in range(D):
            dab = _metric_on(g.dxu[be] if a == be else g.dx[be], I, be, D)
            # Δa = (β == α ? Δ[β][I[β]] : Δu[β][I[β] − 1]),  Δb = (β == α ? Δ[β][I[β] + 1] : Δu[β][I[β]])
            da = _metric_on(g.dx[be], I, be, D) if a == be else _metric_on(g.dxu[be], _shift(I, be, -1), be, D)
            db = _metric_on(g.dx[be], _shift(I, be, +1), be, D) if a == be else _metric_on(g.dxu[be], I, be, D)
            ca = np.where(da > 2 * EPS, 1.0 / da / dab, 0.0)
            cb = np.where(db > 2 * EPS, 1.0 / db / dab, 0.0)
            row = _take(ilin, I, (a,))
            i += [row, row, row]
            j += [_take(ilin, _shift(I, be, -1), (a,)), _take(ilin, _shift(I, be, +1), (a,)), row]
            v += [ca, cb, -(ca + cb)]
    return _coo(i, j, v, (n, n))
