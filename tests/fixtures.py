"""Problem geometries of the reference's own test-suite, rebuilt for the oracle and the HIP path.

  setup2d / setup3d : test/operators.jl:1-49   (16-cell tanh/cosine stretched, all-Dirichlet, Re=1e3)
  setup_mixed       : test/matrices.jl:1-17    (11x7x5, Periodic x Dirichlet/Pressure x Symmetric)
  setup_psolver     : test/psolvers.jl:2-6     (32^2 periodic 2π box)
"""
import numpy as np


def setup2d(o):
    x = (o.tanh_grid(0.0, 1.0, 16), o.tanh_grid(0.0, 1.0, 16, 1.3))
    bc = (o.DirichletBC(), o.DirichletBC())
    return o.make_setup(x, (bc, bc), Re=1000.0)


def setup3d(o):
    x = (o.tanh_grid(0.0, 1.0, 16, 1.2), o.tanh_grid(0.0, 1.0, 16, 1.1), o.cosine_grid(0.0, 1.0, 16))
    bc = (o.DirichletBC(), o.DirichletBC())
    return o.make_setup(x, (bc, bc, bc), Re=1000.0)


def setup_mixed(o):
    x = (o.tanh_grid(0.0, 5.0, 11), o.cosine_grid(0.0, 1.0, 7), o.tanh_grid(0.0, 0.8, 5))
    bcs = (
        (o.PeriodicBC(), o.PeriodicBC()),
        (o.DirichletBC(), o.PressureBC()),
        (o.SymmetricBC(), o.SymmetricBC()),
    )
    return o.make_setup(x, bcs, Re=1000.0)


def setup_psolver(o, n=32):
    x = (np.linspace(0, 2 * np.pi, n + 1), np.linspace(0, 2 * np.pi, n + 1))
    return o.make_setup(x, Re=1000.0)


def setup_periodic(o, n, D=3, L=1.0, Re=1000.0):
    if isinstance(n, int):
        n = (n,) * D
    x = tuple(np.linspace(0.0, L, ni + 1) for ni in n)
    return o.make_setup(x, Re=Re)


def uref(a, x, y, *args):
    """test/operators.jl:22 — same closed form used for 2-D and 3-D."""
    return -(a == 0) * np.sin(x) * np.cos(y) + (a == 1) * np.cos(x) * np.sin(y) + 0 * sum(args, 0.0)


def randn_field(shape, seed):
    rng = np.random.default_rng(seed)
    return np.asfortranarray(rng.standard_normal(shape))


def weighted_inner(o, setup, u, c):
    """Σ_α Σ_{Iu[α]} u·Ωu·c   (test/operators.jl:112-125)."""
    g = setup.grid
    D = g.D
    tot = 0.0
    for a in range(D):
        w = u[..., a] * c[..., a]
        for b in range(D):
            shape = [1] * D
            shape[b] = g.N[b]
            w = w * (g.dxu[b] if a == b else g.dx[b]).reshape(shape)
        tot += float(np.sum(w[tuple(slice(lo, hi) for lo, hi in g.Iu[a])]))
    return tot
