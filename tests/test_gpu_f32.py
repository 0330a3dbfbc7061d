"""The `_f32` entry-point family (T = Float32; docs/src/manual/precision.md:3-16, examples/DecayingTurbulence3D.jl:16) against the
oracle.  The checker is the numpy oracle run on the float32-rounded inputs in float64 — the exact result those inputs define — and the
tolerances are float32 ones: a K1 value is a sum of ~30 products of O(u²/h) magnitude, so errors are measured against max|F| at
a few hundred eps32 (observed ~3e-6); the projection / RK step at 2e-5 relative L2 (fp32 FFT round trip).  fp64 vs fp32 of the same
kernel shows the expected 1e-7-level agreement, which rules out a wrong stencil hiding behind a loose tolerance."""
import numpy as np
import pytest
import torch

from tests import fixtures as fx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ins():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ins_amd

    return ins_amd


def rell2(a, b):
    return float(np.sqrt(np.sum((a - b) ** 2)) / max(np.sqrt(np.sum(b**2)), 1e-300))


def exact_box(o, n, Re=500.0):
    return o.make_setup(tuple(np.arange(ni + 1) * 2.0**-6 for ni in n), Re=Re)


BOXES = [(128, 16, 12), (96, 10, 8), (256, 40, 72), (20, 12, 14), (24, 18)]  # wide 3-D (flux64<float>), narrow 3-D and 2-D (plain kernel)


@pytest.mark.parametrize("n", BOXES)
def test_momentum_f32_matches_oracle(ins, oracle, n):
    o = oracle
    f32 = ins.f32
    so = exact_box(o, n) if len(n) == 3 and n[0] >= 66 else fx.setup_periodic(o, n, D=len(n), Re=500.0)
    D = len(n)
    sp = ins.Setup(x=tuple(so.grid.x[a][1:-1] for a in range(D)), Re=so.Re)
    u32 = fx.randn_field(so.grid.N + (D,), 3).astype(np.float32)
    u_h = o.apply_bc_u(np.asfortranarray(u32.astype(np.float64)), 0.0, so)  # exact ghost copies of the float32 values
    want = o.momentum(u_h, None, 0.0, so)
    u = f32.to_f32(sp, u_h)
    F = f32.vectorfield32(sp)
    F.fill_(7.0)
    got = f32.momentum32_(F, u, sp).cpu().numpy().astype(np.float64)
    ip = tuple(slice(lo, hi) for lo, hi in so.grid.Ip)
    err = np.max(np.abs(got[ip] - want[ip])) / np.max(np.abs(want[ip]))
    assert err < 2e-5, err
    # the fp64 kernel on the same (float32-representable) input: agreement at the float32 rounding level pins the stencil itself
    F64 = ins.to_numpy(ins.momentum(ins.from_numpy(sp, u_h), None, 0.0, sp))
    assert np.max(np.abs(got[ip] - F64[ip])) / np.max(np.abs(F64[ip])) < 2e-5


@pytest.mark.parametrize("n", [(128, 16, 12), (32, 16, 64), (24, 18), (66, 12, 10),  # last: rocFFT non-power-of-two sizes
                               (32, 16, 512), (64, 16, 256), (32, 32, 192), (32, 16, 384)])  # the float2 z pass (three-pass kernel: every length it has)
def test_project_and_poisson_f32_match_oracle(ins, oracle, n):
    o = oracle
    f32 = ins.f32
    D = len(n)
    so = fx.setup_periodic(o, n, D=D)
    sp = ins.Setup(x=tuple(so.grid.x[a][1:-1] for a in range(D)), Re=so.Re)
    pso = o.psolver_spectral(so)
    ps = f32.psolver_spectral32(sp)
    u_h = np.asfortranarray(fx.randn_field(so.grid.N + (D,), 24).astype(np.float32).astype(np.float64))
    u_h = o.apply_bc_u(u_h, 0.0, so)
    want = o.project_(u_h.copy(order="F"), so, pso, o.scalarfield(so))
    o.apply_bc_u_(want, 0.0, so)
    u = f32.to_f32(sp, u_h)
    p = f32.scalarfield32(sp)
    f32.project32_(u, sp, ps, p)
    got = u.cpu().numpy().astype(np.float64)
    assert rell2(got, want) < 2e-5  # ghosts included
    # divergence-free at float32 level: max|div u| h / max|u| ~ eps32 * few
    h = 1.0 / max(n)  # the finest spacing sets the rounding level of a difference quotient
    assert f32.max_abs_divergence32(u, sp, ps) * h < 2e-5 * float(u.abs().max())
    # psolver(p) alone
    f = fx.randn_field(so.grid.N, 23)
    ip = tuple(slice(lo, hi) for lo, hi in so.grid.Ip)
    f[ip] -= f[ip].mean()
    f = np.asfortranarray(f.astype(np.float32).astype(np.float64))
    wantp = o.poisson(pso, f.copy(order="F"))
    gotp = ps(f32.to_f32(sp, f)).cpu().numpy().astype(np.float64)
    assert rell2(gotp[ip], wantp[ip]) < 2e-5
    del ps, sp


@pytest.mark.parametrize("n,method", [((128, 16, 16), "RK44"), ((128, 16, 16), "Wray3"), ((32, 16, 16), "RK44"), ((32, 32), "RK44"), ((128, 16, 16), "FE11")])
def test_rk_step_f32_matches_oracle(ins, oracle, n, method):
    """Three explicit RK steps in float32 (fused stage epilogue on the wide box, plain kernels elsewhere) against the fp64 oracle started
    from the same float32-representable field; and against the library's own fp64 path."""
    o = oracle
    f32 = ins.f32
    D = len(n)
    so = exact_box(o, n) if D == 3 and n[0] >= 66 else fx.setup_periodic(o, n, D=D, Re=500.0)
    sp = ins.Setup(x=tuple(so.grid.x[a][1:-1] for a in range(D)), Re=so.Re)
    pso = o.psolver_spectral(so)
    u0 = o.random_field(so, kp=2, seed=5, psolver=pso)
    u0 = o.apply_bc_u(np.asfortranarray(u0.astype(np.float32).astype(np.float64)), 0.0, so)
    mo, mp_ = getattr(o, method)(), getattr(ins.RKMethods, method)()
    want = o.solve_unsteady(so, (0.0, 0.03), u0, method=mo, psolver=pso, dt=0.01)["u"]
    ps = f32.psolver_spectral32(sp)
    cache = f32.ERKCache32(mp_, sp, ps)
    u = f32.to_f32(sp, u0)
    for _ in range(3):
        f32.timestep32_(cache, u, 0.01)
    got = u.cpu().numpy().astype(np.float64)
    assert rell2(got, want) < 5e-5
    assert f32.max_abs_divergence32(u, sp, ps) / n[0] < 5e-5 * float(u.abs().max())
    del cache, ps, sp


@pytest.mark.parametrize("n,method", [((128, 16, 256), "RK44"), ((256, 16, 192), "Wray3"), ((128, 16, 16), "RK44")])
def test_rk_steps_f32_chained_match_single_steps_and_oracle(ins, oracle, n, method):
    """`timesteps32_` (ins_rk_steps_f32: stage-velocity basis, the correction of every step but the last folded into the next step's first stage kernel —
    on boxes whose z side the float2 solver takes; plain single steps elsewhere) against single `timestep32_` calls, against the k-basis (INS_RK_KEEP_K),
    and against the fp64 oracle at float tolerances."""
    from ins_amd import _lib

    o = oracle
    f32 = ins.f32
    so = exact_box(o, n)
    sp = ins.Setup(x=tuple(so.grid.x[a][1:-1] for a in range(3)), Re=so.Re)
    pso = o.psolver_spectral(so)
    u0 = o.random_field(so, kp=2, seed=5, psolver=pso)
    u0 = o.apply_bc_u(np.asfortranarray(u0.astype(np.float32).astype(np.float64)), 0.0, so)
    mo, mp_ = getattr(o, method)(), getattr(ins.RKMethods, method)()
    want = o.solve_unsteady(so, (0.0, 0.03), u0, method=mo, psolver=pso, dt=0.01)["u"]
    ps = f32.psolver_spectral32(sp)
    cache = f32.ERKCache32(mp_, sp, ps)
    u = f32.timesteps32_(cache, f32.to_f32(sp, u0), 0.01, 3)
    got = u.cpu().numpy().astype(np.float64)
    assert rell2(got, want) < 5e-5
    us = f32.to_f32(sp, u0)
    for _ in range(3):
        f32.timestep32_(cache, us, 0.01)
    assert rell2(us.cpu().numpy().astype(np.float64), got) < 2e-6
    with _lib.options(INS_RK_KEEP_K=1):
        uk = f32.to_f32(sp, u0)
        for _ in range(3):
            f32.timestep32_(cache, uk, 0.01)
    assert rell2(uk.cpu().numpy().astype(np.float64), got) < 2e-6
    assert f32.max_abs_divergence32(u, sp, ps) / n[0] < 5e-5 * float(u.abs().max())
    del cache, ps, sp


def test_f32_family_refuses_other_grids(ins, oracle):
    o = oracle
    f32 = ins.f32
    so = fx.setup3d(o)  # stretched Dirichlet box
    from tests.test_gpu_parity import mirror

    sp = mirror(ins, so, o)
    with pytest.raises(ins.INSHipError, match="periodic uniform"):
        f32.psolver_spectral32(sp)
    with pytest.raises(ins.INSHipError, match="periodic uniform"):
        f32.momentum32_(f32.vectorfield32(sp), f32.vectorfield32(sp), sp)
    with pytest.raises(TypeError):
        f32.momentum32_(ins.vectorfield(sp), ins.vectorfield(sp), sp)
