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


def test_f32_spectral_solver_refuses_other_grids(ins, oracle):
    o = oracle
    f32 = ins.f32
    so = fx.setup3d(o)  # stretched Dirichlet box
    from tests.test_gpu_parity import mirror

    sp = mirror(ins, so, o)
    with pytest.raises(ins.INSHipError, match="periodic uniform"):  # the spectral solver is what stays periodic-only (pressure.jl:289-293)
        f32.psolver_spectral32(sp)
    with pytest.raises(TypeError):
        f32.momentum32_(ins.vectorfield(sp), ins.vectorfield(sp), sp)


# ------------------------------------------------------------------------------------------------------------------------------------
# The family on every other grid of the fp64 family (csrc/ins_f32g.hip): walls, symmetric / pressure sides, stretched spacings, 2-D and 3-D.
# Checker as above: the oracle in float64 on the float32-rounded inputs; float32 tolerances (a stretched 16-cell tanh grid has spacing
# ratios ~30, so the momentum sum is judged against max|F|).
def _lid_setup(o, n=(24, 16, 12)):
    """the shape of examples/LidDrivenCavity3D.jl: cosine x cosine x periodic, moving lid"""
    x = (o.cosine_grid(0.0, 1.0, n[0]), o.cosine_grid(0.0, 1.0, n[1]), np.linspace(-0.2, 0.2, n[2] + 1))
    lid = (o.DirichletBC(), o.DirichletBC((1.0, 0.0, 0.2)))
    return o.make_setup(x, ((o.DirichletBC(), o.DirichletBC()), lid, (o.PeriodicBC(), o.PeriodicBC())), Re=1000.0)


GENERAL = {"dirichlet2d": fx.setup2d, "dirichlet3d": fx.setup3d, "mixed3d": fx.setup_mixed, "lid3d": _lid_setup}


def _mirror(ins, so, o):
    from tests.test_gpu_parity import mirror

    return mirror(ins, so, o)


@pytest.mark.parametrize("geom", list(GENERAL))
def test_general_grid_operators_f32_match_oracle(ins, oracle, geom):
    o, f32 = oracle, ins.f32
    so = GENERAL[geom](o)
    sp = _mirror(ins, so, o)
    g, D = so.grid, so.grid.D
    raw_u = fx.randn_field(g.N + (D,), 4).astype(np.float32)
    raw_p = fx.randn_field(g.N, 5).astype(np.float32)
    # ghost fills are copies / constants: exact in float32
    got = f32.apply_bc_u32_(f32.to_f32(sp, raw_u), sp).cpu().numpy()
    assert np.array_equal(got, o.apply_bc_u(np.asfortranarray(raw_u.astype(np.float64)), 0.0, so).astype(np.float32))
    got = f32.apply_bc_p32_(f32.to_f32(sp, raw_p), sp).cpu().numpy()
    assert np.array_equal(got, o.apply_bc_p(np.asfortranarray(raw_p.astype(np.float64)), 0.0, so).astype(np.float32))
    u_h = o.apply_bc_u(np.asfortranarray(raw_u.astype(np.float64)), 0.0, so)
    want = o.momentum(u_h, None, 0.0, so)
    F = f32.vectorfield32(sp)
    F.fill_(7.0)
    got = f32.momentum32_(F, f32.to_f32(sp, u_h), sp).cpu().numpy().astype(np.float64)
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 2e-5  # whole padded array: zeros outside the degrees of freedom included
    assert np.array_equal(got == 0.0, want == 0.0)


@pytest.mark.parametrize("geom", list(GENERAL))
@pytest.mark.parametrize("solver", ["direct", "cg"])
def test_general_grid_projection_and_poisson_f32_match_oracle(ins, oracle, geom, solver):
    o, f32 = oracle, ins.f32
    so = GENERAL[geom](o)
    sp = _mirror(ins, so, o)
    g, D = so.grid, so.grid.D
    pso = o.psolver_direct(so)
    singular = not any(isinstance(b, o.PressureBC) for side in so.boundary_conditions for b in side)
    # CG on a singular system: the bordered form (mean of the right-hand side removed, as psolver_direct does, pressure.jl:133-140) — a float32-rounded
    # right-hand side is mean-free only to 1e-8
    ps64 = ins.psolver_direct(sp) if solver == "direct" else ins.psolver_cg(sp, abstol=1e-12, reltol=1e-12, bordered=singular)
    ps = f32.psolver_wrap32(sp, ps64)
    u32 = fx.randn_field(g.N + (D,), 24).astype(np.float32)
    u_h = o.apply_bc_u(np.asfortranarray(u32.astype(np.float64)), 0.0, so)
    want = o.apply_bc_u(o.project_(u_h.copy(order="F"), so, pso, o.scalarfield(so)), 0.0, so)
    u = f32.to_f32(sp, u_h)
    p = f32.scalarfield32(sp)
    f32.project32_(u, sp, ps, p)
    f32.apply_bc_u32_(u, sp)
    assert rell2(u.cpu().numpy().astype(np.float64), want) < 2e-5
    # the projected float field is solenoidal at float32 level (relative to the size of one velocity difference over the smallest spacing)
    hmin = min(float(np.min(g.dx[a][1:-1])) for a in range(D))
    assert f32.max_abs_divergence32(u, sp, ps) * hmin < 3e-5 * float(np.max(np.abs(want)))
    # psolver(p) on a solvable right-hand side
    f = fx.randn_field(g.N, 23)
    ip = tuple(slice(lo, hi) for lo, hi in g.Ip)
    if singular:
        f[ip] -= f[ip].mean()
    f = np.asfortranarray(f.astype(np.float32).astype(np.float64))
    want_p = o.poisson(pso, f)
    got_p = ps(f32.to_f32(sp, f)).cpu().numpy().astype(np.float64)
    assert rell2(got_p[ip], want_p[ip]) < 2e-5


@pytest.mark.parametrize("geom", list(GENERAL))
@pytest.mark.parametrize("method", ["RK44", "Wray3", "FE11"])
def test_general_grid_rk_steps_f32_match_oracle(ins, oracle, geom, method):
    """timestep! with T = Float32 on wall-bounded / stretched grids (step_explicit_runge_kutta.jl:4-59): ghost fill, momentum!, stage combination,
    ghost fill, project!, ghost fill — stage by stage on the float kernels; three steps against the oracle in float64."""
    o, f32 = oracle, ins.f32
    so = GENERAL[geom](o)
    sp = _mirror(ins, so, o)
    g, D = so.grid, so.grid.D
    pso = o.psolver_direct(so)
    u0 = 0.5 * fx.randn_field(g.N + (D,), 31).astype(np.float32).astype(np.float64)
    u0 = o.apply_bc_u(o.project_(o.apply_bc_u(np.asfortranarray(u0), 0.0, so), so, pso, o.scalarfield(so)), 0.0, so)
    u0 = np.asfortranarray(u0.astype(np.float32).astype(np.float64))
    dt = 0.3 * o.get_cfl_timestep(u0, so)
    mo = getattr(o, method)()
    st = dict(setup=so, psolver=pso, u=u0.copy(order="F"), t=0.0, n=0)
    oc = o.ode_method_cache(mo, so)
    for _ in range(3):
        st = o.timestep_(mo, st, dt, oc)
    ps = f32.default_psolver32(sp)
    assert isinstance(ps, f32.psolver_wrap32)
    cache = f32.ERKCache32(getattr(ins.RKMethods, method)(), sp, ps)
    u = f32.to_f32(sp, u0)
    f32.timesteps32_(cache, u, dt, 2)
    f32.timestep32_(cache, u, dt)
    assert rell2(u.cpu().numpy().astype(np.float64), st["u"]) < 5e-5
    hmin = min(float(np.min(g.dx[a][1:-1])) for a in range(D))
    assert f32.max_abs_divergence32(u, sp, ps) * hmin < 3e-5 * float(np.max(np.abs(st["u"])))


def test_f32_family_refuses_what_it_does_not_cover(ins, oracle):
    o, f32 = oracle, ins.f32
    x = tuple(np.linspace(0.0, 1.0, 17) for _ in range(2))
    moving = ins.DirichletBC(lambda a, x, y, t: (a == 0) * np.cos(t) + 0 * x)
    sp = ins.Setup(x=x, boundary_conditions=((ins.DirichletBC(), ins.DirichletBC()), (ins.DirichletBC(), moving)), Re=100.0)
    with pytest.raises(NotImplementedError, match="constant boundary data"):
        f32.apply_bc_u32_(f32.vectorfield32(sp), sp)
    # a spectral float solver on a wall-bounded grid: the error names the route that exists
    sw = ins.Setup(x=x, boundary_conditions=((ins.DirichletBC(), ins.DirichletBC()),) * 2, Re=100.0)
    with pytest.raises(Exception, match="ins_poisson_wrap_f32"):
        f32.psolver_spectral32(sw)
    # a Float32 solver handle wraps an fp64 solver of the SAME grid only
    other = ins.Setup(x=tuple(np.linspace(0.0, 1.0, 13) for _ in range(2)), boundary_conditions=((ins.DirichletBC(), ins.DirichletBC()),) * 2, Re=100.0)
    with pytest.raises(ins.INSHipError, match="different grid"):
        f32.psolver_wrap32(sw, ins.psolver_direct(other))
    # and a stepper cache on a wall-bounded grid needs such a handle (there is no spectral float solver to hand it)
    ps = f32.psolver_wrap32(sw)
    cache = f32.ERKCache32(ins.RKMethods.RK44(), sw, ps)
    u = f32.vectorfield32(sw)
    f32.timestep32_(cache, u, 1e-3)
    assert bool(torch.isfinite(u).all())
