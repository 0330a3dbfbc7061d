"""Randomised geometry sweep: small 2-D / 3-D boxes with random sizes, random stretching and random (valid) combinations of Periodic /
Dirichlet / Symmetric / Pressure boundary conditions; every operator of the C ABI against the oracle.  Fixed seeds, so failures reproduce."""
import numpy as np
import pytest

from tests import fixtures as fx
from tests.test_gpu_parity import mirror, relmax

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ins():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ins_amd

    return ins_amd


def random_setup(o, rng):
    D = int(rng.integers(2, 4))
    n = [int(rng.integers(4, 23)) for _ in range(D)]
    if rng.random() < 0.3:
        n[0] = int(rng.integers(60, 140))  # more than one wavefront in x
    x, bcs = [], []
    for a in range(D):
        kind = rng.choice(["periodic", "dirichlet", "symmetric", "pressure", "mixed"])
        L = float(rng.uniform(0.5, 3.0))
        if kind == "periodic":
            bcs.append((o.PeriodicBC(), o.PeriodicBC()))
            x.append(np.linspace(0.0, L, n[a] + 1) if rng.random() < 0.7 else o.tanh_grid(0.0, L, n[a], 1.1))
        else:
            pool = {"dirichlet": [o.DirichletBC], "symmetric": [o.SymmetricBC], "pressure": [o.PressureBC],
                    "mixed": [o.DirichletBC, o.SymmetricBC, o.PressureBC]}[kind]
            mk = lambda: (lambda c: c((0.3, -0.2, 0.1)[:D]) if c is o.DirichletBC and rng.random() < 0.5 else c())(pool[int(rng.integers(len(pool)))])
            bcs.append((mk(), mk()))
            g = rng.random()
            x.append(o.cosine_grid(0.0, L, n[a]) if g < 0.4 else (o.tanh_grid(0.0, L, n[a], 1.3) if g < 0.8 else np.linspace(0.0, L, n[a] + 1)))
    return o.make_setup(tuple(x), tuple(bcs), Re=float(rng.uniform(50, 2000)))


@pytest.mark.parametrize("seed", range(24))
def test_random_geometry(ins, oracle, seed):
    o = oracle
    rng = np.random.default_rng(1000 + seed)
    so = random_setup(o, rng)
    sp = mirror(ins, so, o)
    g = so.grid
    D = g.D
    tol = 2e-12
    u_raw, p_raw = fx.randn_field(g.N + (D,), seed), fx.randn_field(g.N, seed + 1)
    u_h, p_h = o.apply_bc_u(u_raw, 0.0, so), o.apply_bc_p(p_raw, 0.0, so)
    u_d = ins.apply_bc_u(ins.from_numpy(sp, u_raw), 0.0, sp)
    p_d = ins.apply_bc_p(ins.from_numpy(sp, p_raw), 0.0, sp)
    assert np.array_equal(ins.to_numpy(u_d), u_h) and np.array_equal(ins.to_numpy(p_d), p_h)
    checks = {
        "momentum": (ins.momentum(u_d, None, 0.0, sp), o.momentum(u_h, None, 0.0, so)),
        "convection": (ins.convection(u_d, sp), o.convection(u_h, so)),
        "diffusion": (ins.diffusion(u_d, sp), o.diffusion(u_h, so)),
        "divergence": (ins.divergence(u_d, sp), o.divergence(u_h, so)),
        "pressuregradient": (ins.pressuregradient(p_d, sp), o.pressuregradient(p_h, so)),
        "laplacian": (ins.laplacian(p_d, sp), o.laplacian(p_h, so)),
        "vorticity": (ins.vorticity(u_d, sp), o.vorticity(u_h, so)),
        "interpolate_u_p": (ins.interpolate_u_p(u_d, sp), o.interpolate_u_p(u_h, so)),
        "Qfield": (ins.Qfield(u_d, sp), o.Qfield(u_h, so)),
        "strain": (ins.dissipation_from_strain(u_d, sp), o.dissipation_from_strain(u_h, so)),
        "kinetic_energy": (ins.kinetic_energy(u_d, sp), o.kinetic_energy_(o.scalarfield(so), u_h, so)),
    }
    for name, (got, want) in checks.items():
        assert relmax(ins.to_numpy(got), want) < tol, name
    s_h = o.smagorinsky_closure(so)(u_h, 0.13)
    assert relmax(ins.to_numpy(ins.smagorinsky_closure(sp)(u_d, 0.13)), s_h) < 1e-10
    # projection with the default solver (spectral / direct): divergence-free and equal to the oracle's
    ps_h, ps_d = o.default_psolver(so), ins.default_psolver(sp)
    q_h = o.project(u_h, so, ps_h)
    q_d = ins.project(u_d, sp, ps_d)
    scale = max(np.abs(q_h).max(), 1e-300)
    assert np.abs(ins.to_numpy(q_d) - q_h).max() / scale < 1e-9
    # the in-place twin is ONE fused call on the device (divergence, solve, ghost pressures and gradient-subtract inside the solver's passes)
    q2 = ins.project_(ins.copyfield(u_d), sp, ps_d, ins.scalarfield(sp))
    dof = np.zeros(g.N + (D,), dtype=bool)
    for a in range(D):
        dof[tuple(slice(lo, hi) for lo, hi in g.Iu[a]) + (a,)] = True
    assert np.abs(ins.to_numpy(q2) - q_h)[dof].max() / scale < 1e-9
    # one RK step
    o.apply_bc_u_(q_h, 0.0, so)
    ref = o.solve_unsteady(so, (0.0, 1e-3), 0.05 * q_h, psolver=ps_h, dt=1e-3)
    (v, _, _), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 1e-3), ustart=ins.from_numpy(sp, 0.05 * q_h), psolver=ps_d, Δt=1e-3)
    assert np.abs(ins.to_numpy(v) - ref["u"]).max() / max(np.abs(ref["u"]).max(), 1e-300) < 1e-9
