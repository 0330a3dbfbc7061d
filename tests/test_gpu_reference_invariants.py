"""The reference's own operator tests (test/operators.jl:50-230), applied to the HIP operators directly — same geometries
(`Setup2D` / `Setup3D`: 16-cell tanh / cosine stretched, all-Dirichlet, the closed-form field of :22), same invariants, same tolerances:
D = -Gᵀ, negative semi-definite Laplacian equal to `laplacian_mat`, skew-symmetric convection (1e-12), dissipative diffusion,
fused = unfused, and "returns a finite array" for momentum / body force / pressure / the other fields / turbulence statistics."""
import numpy as np
import pytest

from tests import fixtures as fx
from tests.test_gpu_parity import mirror

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ins():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ins_amd

    return ins_amd


def both(ins, o):
    for mk in (fx.setup2d, fx.setup3d):
        so = mk(o)
        sp = mirror(ins, so, o)
        u = ins.velocityfield(sp, fx.uref, 0.0, psolver=ins.default_psolver(sp))  # test/operators.jl:22-24
        yield so, sp, u


def scalar_inner(so, p, q):
    g = so.grid
    w = p * q
    for b in range(g.D):
        shape = [1] * g.D
        shape[b] = g.N[b]
        w = w * g.dx[b].reshape(shape)
    return float(np.sum(w[tuple(slice(lo, hi) for lo, hi in g.Ip)]))


def test_divergence_is_finite(ins, oracle):
    for so, sp, u in both(ins, oracle):
        assert np.isfinite(ins.to_numpy(ins.divergence(u, sp))).all()


def test_pressure_gradient_is_minus_divergence_transposed(ins, oracle):
    o = oracle
    for so, sp, _ in both(ins, o):
        g = so.grid
        v = ins.apply_bc_u(ins.from_numpy(sp, fx.randn_field(g.N + (g.D,), 1)), 0.0, sp)
        p = ins.apply_bc_p(ins.from_numpy(sp, fx.randn_field(g.N, 2)), 0.0, sp)
        Dv, Gp = ins.to_numpy(ins.divergence(v, sp)), ins.to_numpy(ins.pressuregradient(p, sp))
        pDv = scalar_inner(so, ins.to_numpy(p), Dv)
        vGp = fx.weighted_inner(o, so, ins.to_numpy(v), Gp)
        assert pDv == pytest.approx(-vGp, rel=1e-12)


def test_laplacian_is_negative_and_equals_the_matrix(ins, oracle):
    o = oracle
    for so, sp, _ in both(ins, o):
        g = so.grid
        p = ins.apply_bc_p(ins.from_numpy(sp, fx.randn_field(g.N, 3)), 0.0, sp)
        Lp = ins.to_numpy(ins.laplacian(p, sp))
        ph = ins.to_numpy(p)
        sl = tuple(slice(lo, hi) for lo, hi in g.Ip)
        assert float(np.sum((ph * Lp)[sl])) <= 0  # laplacian! already carries the volume (operators.jl:297-364)
        L = o.laplacian_mat(so)
        assert np.sum((Lp[sl].reshape(-1, order="F") - L @ ph[sl].reshape(-1, order="F")) ** 2) < 1e-12


def test_convection_is_skew_symmetric_and_diffusion_dissipative(ins, oracle):
    o = oracle
    for so, sp, u in both(ins, o):
        uh = ins.to_numpy(u)
        assert abs(fx.weighted_inner(o, so, uh, ins.to_numpy(ins.convection(u, sp)))) < 1e-12
        assert fx.weighted_inner(o, so, uh, ins.to_numpy(ins.diffusion(u, sp))) <= 0
        cd = ins.to_numpy(ins.convectiondiffusion_(ins.vectorfield(sp), u, sp))
        assert np.allclose(cd, ins.to_numpy(ins.convection(u, sp)) + ins.to_numpy(ins.diffusion(u, sp)), rtol=1e-12, atol=1e-14)


def test_momentum_bodyforce_pressure_and_other_fields_are_finite(ins, oracle):
    o = oracle
    for so, sp, u in both(ins, o):
        D = so.grid.D
        ps = ins.default_psolver(sp)
        assert np.isfinite(ins.to_numpy(ins.momentum(u, None, 1.0, sp))).all()
        p = ins.pressure(u, None, 0.0, sp, ps)
        assert np.isfinite(ins.to_numpy(p)).all()
        w = ins.vorticity(u, sp)
        assert tuple(w.shape) == (so.grid.N if D == 2 else so.grid.N + (3,))
        fields = [w, ins.smagorinsky_closure(sp)(u, 0.1), ins.interpolate_u_p(u, sp), ins.interpolate_ω_p(w, sp), ins.Qfield(u, sp),
                  ins.kinetic_energy(u, sp), ins.dissipation_from_strain(u, sp)]
        if D == 3:
            fields.append(ins.eig2field(u, sp))
        for f in fields:
            assert np.isfinite(ins.to_numpy(f)).all()
        assert np.isfinite(ins.total_kinetic_energy(u, sp))
        # body force through the setup (test/operators.jl:171-178)
        xin = [so.grid.x[a][1:-1] for a in range(D)]
        sb = ins.Setup(x=xin, boundary_conditions=sp.boundary_conditions, Re=so.Re, bodyforce=lambda a, x, y, *zt: (a == 0) * (1 + 0 * (x + y)) + 0 * sum(zt[:-1], 0.0),
                       issteadybodyforce=False)
        assert np.isfinite(ins.to_numpy(ins.applybodyforce(u, 0.0, sb))).all()


def test_turbulence_statistics(ins, oracle):
    """test/operators.jl:222-230: scale numbers of a random field on a periodic box."""
    for D, n in ((2, 64), (3, 32)):
        sp = ins.Setup(x=(np.linspace(0.0, 1.0, n + 1),) * D, Re=1e4)
        u = ins.random_field(sp, 0.0, kp=5)
        s = ins.get_scale_numbers(u, sp)
        assert all(np.isfinite(v) and v > 0 for v in s.values()) and set(s) == {"uavg", "ϵ", "η", "λ", "Reλ", "L", "τ", "Re_int"}
