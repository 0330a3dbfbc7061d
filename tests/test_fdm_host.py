"""psolver_direct, host part (no GPU): the separable form Σα Tα ⊗ (⊗β≠α Dβ) that the fast-diagonalisation solver
inverts IS `laplacian_mat` (matrices.jl:484-492, oracle restatement), for every boundary-condition mix the reference tests."""
import functools
from types import SimpleNamespace

import numpy as np
import pytest

from tests import fixtures as fx
import ins_amd
from ins_amd.pressure import _laplacian_1d
from oracle import ins_oracle as o


def _channel(o):  # Periodic x Dirichlet (examples/PlaneJets2D.jl-like) with a Symmetric/Pressure pair on z
    x = (np.linspace(0.0, 2.0, 9), o.tanh_grid(0.0, 1.0, 6, 1.5), o.stretched_grid(0.0, 1.0, 5, 1.2))
    bcs = ((o.PeriodicBC(), o.PeriodicBC()), (o.DirichletBC(), o.DirichletBC()), (o.SymmetricBC(), o.PressureBC()))
    return o.make_setup(x, bcs, Re=100.0)


def _pressure_left(o):
    x = (o.cosine_grid(0.0, 1.0, 7), np.linspace(0.0, 1.0, 6))
    bcs = ((o.PressureBC(), o.DirichletBC()), (o.PressureBC(), o.PressureBC()))
    return o.make_setup(x, bcs, Re=100.0)


CASES = {
    "dirichlet2d": fx.setup2d,
    "dirichlet3d": fx.setup3d,
    "mixed3d": fx.setup_mixed,
    "channel3d": _channel,
    "pressure2d": _pressure_left,
}


def host_setup(so):
    """The package-side view (`setup.grid.Ip/Δ/Δu`, `setup.boundary_conditions`) of an oracle setup, without a device."""
    bcs = tuple(tuple(getattr(ins_amd, type(b).__name__)() for b in pair) for pair in so.boundary_conditions)
    grid = SimpleNamespace(Ip=so.grid.Ip, Δ=so.grid.dx, Δu=so.grid.dxu, dimension=so.grid.D)
    return SimpleNamespace(grid=grid, boundary_conditions=bcs)


@pytest.mark.parametrize("name", list(CASES))
def test_kron_sum_is_laplacian_mat(name):
    so = CASES[name](o)
    hs = host_setup(so)
    D = so.grid.D
    T = [_laplacian_1d(hs, a) for a in range(D)]
    Dm = [np.diag(so.grid.dx[a][slice(*so.grid.Ip[a])]) for a in range(D)]
    # column-major (x fastest) DOF order: kron(z, y, x)
    L = sum(functools.reduce(np.kron, [T[b] if b == a else Dm[b] for b in reversed(range(D))]) for a in range(D))
    Lref = o.laplacian_mat(so, dense=True)
    assert L.shape == Lref.shape
    assert np.abs(L - Lref).max() <= 1e-12 * np.abs(Lref).max()


@pytest.mark.parametrize("name", list(CASES))
def test_fdm_solve_matches_sparse_direct(name):
    """Numpy model of csrc/ins_fdm.hip's solve (same eigenpairs, same order of operations) vs. the oracle's sparse LU."""
    so = CASES[name](o)
    hs = host_setup(so)
    D, Np = so.grid.D, so.grid.Np
    V, lam = [], []
    for a in range(D):
        dm = 1.0 / np.sqrt(so.grid.dx[a][slice(*so.grid.Ip[a])])
        l, W = np.linalg.eigh(dm[:, None] * _laplacian_1d(hs, a) * dm[None, :])
        V.append(dm[:, None] * W)
        lam.append(l)
    singular = not any(isinstance(b, o.PressureBC) for pair in so.boundary_conditions for b in pair)
    rng = np.random.default_rng(5)
    f = rng.standard_normal(Np)
    p = o.scalarfield(so)
    p[o._sl(so.grid.Ip)] = f
    pref = o.psolver_direct(so)(p.copy())[o._sl(so.grid.Ip)]
    q = f - f.mean() if singular else f.copy()
    for a in range(D):
        q = np.moveaxis(np.tensordot(V[a].T, q, axes=(1, a)), 0, a)
    den = functools.reduce(np.add.outer, lam)
    tol = 1e-10 * D * max(np.abs(l).max() for l in lam) if singular else 0.0
    q = np.where(np.abs(den) <= tol, 0.0, q / np.where(np.abs(den) <= tol, 1.0, den))
    for a in range(D):
        q = np.moveaxis(np.tensordot(V[a], q, axes=(1, a)), 0, a)
    if singular:
        q = q - q.mean()
    assert np.abs(q - pref).max() <= 1e-10 * np.abs(pref).max()


def _factor_1d(x):
    so = o.make_setup((x, np.linspace(0.0, 1.0, 5)), ((o.DirichletBC(), o.DirichletBC()), (o.PeriodicBC(), o.PeriodicBC())), Re=100.0)
    hs = host_setup(so)
    lo, hi = so.grid.Ip[0]
    dm = 1.0 / np.sqrt(so.grid.dx[0][lo:hi])
    return dm[:, None] * _laplacian_1d(hs, 0) * dm[None, :]


@pytest.mark.parametrize("n", [64, 256, 1024])
def test_even_odd_eigenpairs_only_when_they_satisfy_the_given_matrix(n):
    """The half-size (even / odd) eigen-decomposition is taken for mirror-symmetric grids — the reference's cosine grid is one up to the rounding of its
    coordinates — and its pairs must satisfy the GIVEN 1-D factor at the eigensolver's own rounding level; a grid that is only approximately symmetric
    (wall cells 1e-9 apart) gets LAPACK's pairs of the given matrix instead of pairs of a perturbed one (advisor finding, round 2)."""
    from ins_amd.pressure import _eigh_1d, _sym_slack

    S = _factor_1d(o.cosine_grid(0.0, 1.0, n))
    lam, W = _eigh_1d(S)
    h = n // 2
    assert np.array_equal(W[:h, :h], W[h:, :h][::-1]) and np.array_equal(W[:h, h:], -W[h:, h:][::-1])  # exactly even / odd: the fold applies
    assert np.abs(S @ W - W * lam[None, :]).max() <= _sym_slack(S)
    x = o.cosine_grid(0.0, 1.0, n).copy()
    x[1] *= 1.0 + 1e-9  # first cell 1e-9 (relative) wider than the last
    S2 = _factor_1d(x)
    lam2, W2 = _eigh_1d(S2)
    assert not np.array_equal(W2[:h, :h], W2[h:, :h][::-1])
    assert np.abs(S2 @ W2 - W2 * lam2[None, :]).max() <= _sym_slack(S2)
