"""Grid generators and the `Grid` named tuple (grid.jl), host side.  The O(N) metric vectors are built
with numpy on the host exactly as the reference builds them on the CPU before `adapt(backend, ...)`
(grid.jl:263); the device copies live inside the `ins_grid_t` handle."""
from types import SimpleNamespace

import numpy as np

from .boundary_conditions import offset_p, offset_u, padghost_

EPS = float(np.finfo(np.float64).eps)


def cosine_grid(a, b, N):
    """grid.jl:39-43"""
    i = np.arange(N + 1, dtype=np.float64)
    return a + (b - a) * (1 - np.cos(np.pi * (i / N))) / 2


def stretched_grid(a, b, N, s=1.0):
    """grid.jl:60-67"""
    if s <= 0:
        raise ValueError("The stretch factor must be positive")
    if np.isclose(s, 1.0):
        return np.linspace(a, b, N + 1)
    i = np.arange(N + 1, dtype=np.float64)
    return a + (b - a) * (1 - s**i) / (1 - s**N)


def tanh_grid(a, b, N, gamma=1.0):
    """grid.jl:73-77"""
    x = np.linspace(0.0, 1.0, N + 1)
    return a + (b - a) * (1 + np.tanh(gamma * (2 * x - 1)) / np.tanh(gamma)) / 2


def max_size(grid):
    """grid.jl:22-26"""
    return float(np.sqrt(sum(np.max(d) ** 2 for d in grid.Δ)))


def Grid(x, boundary_conditions):
    """grid.jl:100-276.  Index ranges are 0-based half-open `(lo, hi)` pairs."""
    xs = [[float(v) for v in np.asarray(xi, dtype=np.float64)] for xi in x]
    xlims = tuple((min(xi), max(xi)) for xi in xs)
    D = len(xs)
    if D not in (2, 3):
        raise ValueError("only 2-D and 3-D grids are supported")
    for d in range(D):
        a, b = boundary_conditions[d]
        padghost_(a, xs[d], False)
        padghost_(b, xs[d], True)
    xs = tuple(np.array(xi, dtype=np.float64) for xi in xs)
    N = tuple(len(xi) - 1 for xi in xs)
    bcs = boundary_conditions
    Iu = tuple(
        tuple(
            (offset_u(bcs[be][0], False, al == be), N[be] - offset_u(bcs[be][1], True, al == be))
            for be in range(D)
        )
        for al in range(D)
    )
    Ip = tuple((offset_p(bcs[be][0], False), N[be] - offset_p(bcs[be][1], True)) for be in range(D))
    Nu = tuple(tuple(hi - lo for lo, hi in Iu[al]) for al in range(D))
    Np = tuple(hi - lo for lo, hi in Ip)
    xp = tuple((xi[:-1] + xi[1:]) / 2 for xi in xs)
    xu = tuple(tuple(xs[be][1:] if al == be else xp[be] for be in range(D)) for al in range(D))
    dx = tuple(np.maximum(np.diff(xi), EPS) for xi in xs)
    dxu = tuple(np.maximum(np.append(np.diff(xp[d]), dx[d][-1] / 2), EPS) for d in range(D))

    def weights(al, be):
        if al == be:
            A1 = np.full(N[al], 0.5)
            A1[0] = 1.0
            A2 = np.full(N[al], 0.5)
            A2[-1] = 1.0
        else:
            inner = (xs[be][1 : N[be]] - xp[be][: N[be] - 1]) / dxu[be][: N[be] - 1]
            A1 = np.concatenate([[1.0], 1.0 - inner])
            A2 = np.concatenate([inner, [1.0]])
        return (np.ascontiguousarray(A1), np.ascontiguousarray(A2))

    A = tuple(tuple(weights(al, be) for be in range(D)) for al in range(D))
    return SimpleNamespace(
        xlims=xlims, dimension=D, N=N, Nu=Nu, Np=Np, Iu=Iu, Ip=Ip, x=xs, xu=xu, xp=xp, Δ=dx, Δu=dxu, A=A
    )
