"""Pin the CPU oracle against the reference's OWN known-answer tests and invariants.

The reference ships no golden vectors; these are its test-suite's checks, re-run
against `oracle/ins_oracle.py`.  Each test cites the reference test it restates.
"""
import math

import numpy as np
import pytest

from tests import fixtures as fx


def _ip(setup):
    return tuple(slice(lo, hi) for lo, hi in setup.grid.Ip)


# ------------------------------------------------------------------ test/grid.jl:1-43
def test_grid_generators(oracle):
    o = oracle
    for x in (o.cosine_grid(0.0, 1.0, 10), o.stretched_grid(0.0, 1.0, 10, 1.1), o.tanh_grid(0.0, 1.0, 10, 1.5)):
        assert x[0] == pytest.approx(0.0, abs=1e-15) and x[-1] == pytest.approx(1.0, abs=1e-15)
        assert np.all(np.diff(x) > 0) and len(x) == 11
    assert np.allclose(o.stretched_grid(0.0, 2.0, 4, 1.0), np.linspace(0, 2, 5))


def test_grid_layout_periodic_dirichlet(oracle):
    """SURVEY Appendix A.1 table (boundary_conditions.jl:42-89, grid.jl:114-187)."""
    o = oracle
    s = fx.setup_periodic(o, 8, D=2)
    g = s.grid
    assert g.N == (10, 10) and g.Ip == ((1, 9), (1, 9)) and g.Iu[0] == ((1, 9), (1, 9))
    assert np.allclose(g.dx[0], 1 / 8) and np.allclose(g.dxu[0][:-1], 1 / 8) and g.dxu[0][-1] == pytest.approx(1 / 16)
    assert g.A[0][0][0][0] == 1.0 and g.A[0][0][1][-1] == 1.0 and np.allclose(g.A[0][1][1][:-1], 0.5)
    s = fx.setup2d(o)
    g = s.grid
    assert g.N == (18, 18) and g.Ip == ((1, 17), (1, 17))
    assert g.Iu[0] == ((1, 16), (1, 17)) and g.Iu[1] == ((1, 17), (1, 16))
    assert g.dx[0][0] == o.EPS and g.dx[0][-1] == o.EPS
    assert g.dxu[0][0] == pytest.approx(g.dx[0][1] / 2)
    s = fx.setup_mixed(o)
    g = s.grid
    assert g.N == (13, 9, 7)  # n + 2 per dim (a Pressure BC on the *left* would add a third ghost)
    assert g.Ip[1] == (1, 8) and g.Iu[1][1] == (1, 8) and g.Iu[0][1] == (1, 8)
    assert g.Iu[2][2] == (1, 5) and g.Iu[0][2] == (1, 6)  # Symmetric: normal loses the right face


# ------------------------------------------------------------------ test/operators.jl:51-56
@pytest.mark.parametrize("mk", [fx.setup2d, fx.setup3d])
def test_divergence_finite(oracle, mk):
    o = oracle
    s = mk(o)
    u = o.velocityfield(s, fx.uref, 0.0)
    assert np.all(np.isfinite(o.divergence(u, s)))


# ------------------------------------------------------------------ test/operators.jl:58-88
@pytest.mark.parametrize("mk", [fx.setup2d, fx.setup3d, fx.setup_mixed])
def test_pressuregradient_is_minus_divergence_transpose(oracle, mk):
    o = oracle
    s = mk(o)
    g = s.grid
    v = o.apply_bc_u(fx.randn_field(g.N + (g.D,), 1), 0.0, s)
    p = o.apply_bc_p(fx.randn_field(g.N, 2), 0.0, s)
    if mk is fx.setup_mixed:
        pytest.skip("adjointness is only asserted on the all-Dirichlet fixtures in the reference")
    Dv = o.scalewithvolume(o.divergence(v, s), s)
    Gp = o.pressuregradient(p, s)
    pDv = float(np.sum((p * Dv)[_ip(s)]))
    vGp = fx.weighted_inner(o, s, v, Gp)
    assert pDv == pytest.approx(-vGp, rel=math.sqrt(o.EPS))


# ------------------------------------------------------------------ test/operators.jl:90-105
@pytest.mark.parametrize("mk", [fx.setup2d, fx.setup3d, fx.setup_mixed])
def test_laplacian_matches_matrix_and_is_negative(oracle, mk):
    o = oracle
    s = mk(o)
    g = s.grid
    p = o.apply_bc_p(fx.randn_field(g.N, 3), 0.0, s)
    Lp = o.laplacian(p, s)
    OLp = o.scalewithvolume(Lp, s)
    assert float(np.sum((p * OLp)[_ip(s)])) <= 0
    Lmat_p = o.laplacian_mat_apply(p[_ip(s)].reshape(-1, order="F"), s)
    assert float(np.sum((Lp[_ip(s)].reshape(-1, order="F") - Lmat_p) ** 2)) == pytest.approx(0, abs=1e-12)


def test_laplacian_mat_symmetric_nsd(oracle):
    """pressure.jl:137 'Matrix not symmetric' guard; spatial.md:203-216."""
    o = oracle
    s = o.make_setup((o.tanh_grid(0.0, 1.0, 6), o.cosine_grid(0.0, 1.0, 5)), ((o.DirichletBC(),) * 2,) * 2)
    L = o.laplacian_mat(s, dense=True)
    assert np.allclose(L, o.laplacian_mat_dense_probe(s), rtol=1e-13, atol=1e-13)
    sm = fx.setup_mixed(o)
    assert np.allclose(o.laplacian_mat(sm, dense=True), o.laplacian_mat_dense_probe(sm), rtol=1e-13, atol=1e-13)
    sp_ = fx.setup_periodic(o, (6, 8), D=2)
    assert np.allclose(o.laplacian_mat(sp_, dense=True), o.laplacian_mat_dense_probe(sp_), rtol=1e-13, atol=1e-13)
    assert np.max(np.abs(L - L.T)) < 1e-12
    w = np.linalg.eigvalsh((L + L.T) / 2)
    assert w.max() < 1e-10
    assert np.max(np.abs(L @ np.ones(L.shape[0]))) < 1e-10  # constants in the null space


# ------------------------------------------------------------------ test/operators.jl:107-128
@pytest.mark.parametrize("mk", [fx.setup2d, fx.setup3d])
def test_convection_skew_symmetric(oracle, mk):
    o = oracle
    s = mk(o)
    u = o.velocityfield(s, fx.uref, 0.0)
    c = o.convection(u, s)
    assert abs(fx.weighted_inner(o, s, u, c)) < 1e-12


def test_convection_skew_symmetric_periodic(oracle):
    o = oracle
    s = fx.setup_periodic(o, 12, D=3)
    u = o.random_field(s, kp=3, seed=5)
    c = o.convection(u, s)
    assert abs(fx.weighted_inner(o, s, u, c)) < 1e-12 * max(1.0, fx.weighted_inner(o, s, u, u))


# ------------------------------------------------------------------ test/operators.jl:130-151
@pytest.mark.parametrize("mk", [fx.setup2d, fx.setup3d])
def test_diffusion_dissipative(oracle, mk):
    o = oracle
    s = mk(o)
    u = o.velocityfield(s, fx.uref, 0.0)
    d = o.diffusion(u, s)
    assert fx.weighted_inner(o, s, u, d) <= 0


# ------------------------------------------------------------------ test/operators.jl:153-160
@pytest.mark.parametrize("mk", [fx.setup2d, fx.setup3d, fx.setup_mixed])
def test_fused_equals_unfused(oracle, mk):
    o = oracle
    s = mk(o)
    g = s.grid
    if mk is fx.setup_mixed:
        u = o.apply_bc_u(fx.randn_field(g.N + (g.D,), 7), 0.0, s)
    else:
        u = o.velocityfield(s, fx.uref, 0.0)
    cd = o.convectiondiffusion_(o.vectorfield(s), u, s)
    c = o.convection(u, s)
    d = o.diffusion(u, s)
    assert np.allclose(cd, c + d, rtol=math.sqrt(o.EPS), atol=0)
    assert np.all(np.isfinite(cd))


# ------------------------------------------------------------------ test/psolvers.jl:1-32
def test_pressure_solvers_known_answer(oracle):
    o = oracle
    s = fx.setup_psolver(o, 32)
    g = s.grid
    X, Y = g.xp[0].reshape(-1, 1), g.xp[1].reshape(1, -1)
    p_exact = np.asfortranarray(0.25 * (np.cos(2 * X) + np.cos(2 * Y)))
    o.apply_bc_p_(p_exact, 0.0, s)
    lap = o.laplacian(p_exact, s)
    for mk in (o.psolver_direct, o.psolver_cg, o.psolver_spectral):
        got = o.apply_bc_p(o.poisson(mk(s), lap), 0.0, s)
        assert np.allclose(got[_ip(s)], p_exact[_ip(s)], rtol=math.sqrt(o.EPS), atol=1e-9)


def test_cg_dirichlet_matches_direct(oracle):
    """psolver_cg vs psolver_direct on the non-periodic fixture (default_psolver's branch, pressure.jl:93-97)."""
    o = oracle
    s = o.make_setup((o.tanh_grid(0.0, 1.0, 8), o.cosine_grid(0.0, 1.0, 6)), ((o.DirichletBC(),) * 2,) * 2)
    u = o.apply_bc_u(fx.randn_field(s.grid.N + (2,), 11), 0.0, s)
    f = o.scalewithvolume(o.divergence(u, s), s)
    pd = o.poisson(o.psolver_direct(s), f)
    pc = o.poisson(o.psolver_cg(s, reltol=1e-13), f)
    ip = _ip(s)
    a, b = pd[ip] - pd[ip].mean(), pc[ip] - pc[ip].mean()
    assert np.allclose(a, b, rtol=1e-7, atol=1e-9)


# ------------------------------------------------------------------ methods.jl:177-182
@pytest.mark.parametrize("D", [2, 3])
def test_projection_gives_divergence_free(oracle, D):
    o = oracle
    s = fx.setup_periodic(o, 16, D=D)
    g = s.grid
    u = o.apply_bc_u(fx.randn_field(g.N + (D,), 4), 0.0, s)
    ps = o.psolver_spectral(s)
    v = o.project(u, s, ps)
    o.apply_bc_u_(v, 0.0, s)
    assert np.max(np.abs(o.divergence(v, s)[_ip(s)])) < 1e-11
    # in-place twin (pressure.jl:69-82) == out-of-place (pressure.jl:52-66)
    w = u.copy(order="F")
    o.project_(w, s, ps, o.scalarfield(s))
    o.apply_bc_u_(w, 0.0, s)
    assert np.allclose(w, v, rtol=1e-12, atol=1e-13)


# ------------------------------------------------------------------ RK tableau shift, methods.jl:231-236
def test_rk44_shifted_tableau(oracle):
    m = oracle.RK44()
    assert np.allclose(m.A, [[0.5, 0, 0, 0], [0, 0.5, 0, 0], [0, 0, 1, 0], [1 / 6, 1 / 3, 1 / 3, 1 / 6]])
    assert np.allclose(m.c, [0.5, 0.5, 1, 1])


# ------------------------------------------------------------------ examples/TaylorGreenVortex2D.jl:29-74
def _tgv2d_error(o, n, tend, dt):
    Re = 2000.0
    x = (np.linspace(0, 2 * np.pi, n + 1),) * 2
    s = o.make_setup(x, Re=Re)
    ps = o.psolver_spectral(s)
    sol = o.tgv2d_ufunc(Re)
    u0 = o.velocityfield(s, sol(0.0), 0.0, psolver=ps)
    ut = o.velocityfield(s, sol(tend), tend, psolver=ps, doproject=False)
    st = o.solve_unsteady(s, (0.0, tend), u0, psolver=ps, dt=dt)
    ip = _ip(s)
    err = math.sqrt(float(np.sum((st["u"][ip] - ut[ip]) ** 2))) / math.sqrt(float(np.sum(ut[ip] ** 2)))
    div = float(np.max(np.abs(o.divergence(st["u"], s)[ip])))
    return err, div


def test_tgv2d_second_order_convergence(oracle):
    errs = []
    for n in (8, 16, 32):
        e, div = _tgv2d_error(oracle, n, 0.5, 0.01)
        assert div < 1e-13
        errs.append(e)
    assert 3.5 < errs[0] / errs[1] < 4.5 and 3.5 < errs[1] / errs[2] < 4.5
    # Regression anchors from SURVEY.md Appendix A.6 (an independent transcription of the reference)
    assert errs[0] == pytest.approx(2.518e-5, rel=2e-3)
    assert errs[1] == pytest.approx(6.393e-6, rel=2e-3)
    assert errs[2] == pytest.approx(1.604e-6, rel=2e-3)


def test_tgv2d_config1_128(oracle):
    """BASELINE config 1: TGV2D 128² fp64, RK4 + spectral Poisson (short horizon to stay fast)."""
    e, div = _tgv2d_error(oracle, 128, 0.1, 0.01)
    assert e < 5e-7 and div < 1e-12


# ------------------------------------------------------------------ test/timesteppers.jl (stepper consistency)
def test_timestep_energy_decays_and_stays_solenoidal(oracle):
    o = oracle
    s = fx.setup_periodic(o, 16, D=3)
    ps = o.psolver_spectral(s)
    u = o.velocityfield(s, o.tgv3d_ufunc, 0.0, psolver=ps)
    e0 = o.total_kinetic_energy(u, s)
    st = o.solve_unsteady(s, (0.0, 0.02), u, psolver=ps, dt=0.01)
    assert st["n"] == 2 and st["t"] == pytest.approx(0.02)
    assert o.total_kinetic_energy(st["u"], s) < e0
    assert np.max(np.abs(o.divergence(st["u"], s)[_ip(s)])) < 1e-12


def test_cfl_timestep(oracle):
    o = oracle
    s = fx.setup_periodic(o, 16, D=2)
    u = o.vectorfield(s)
    u[..., 0] = 2.0
    dt = o.get_cfl_timestep(u, s)
    assert dt == pytest.approx(min(s.Re * (1 / 16) ** 2 / 2, (1 / 16) / 2.0))


def test_c_port_matches_numpy_oracle(oracle):
    """The C/OpenMP restatement (oracle/c, bench.py's CPU baseline) against the numpy oracle: 2 RK44 steps."""
    import subprocess, os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle", "c")], stdout=subprocess.DEVNULL)
    from oracle.c_port import CPort

    o = oracle
    s = fx.setup_periodic(o, (12, 10, 14), D=3, Re=300.0)
    ps = o.psolver_spectral(s)
    u0 = o.random_field(s, kp=2, seed=4, psolver=ps)
    st = o.solve_unsteady(s, (0.0, 0.02), u0, psolver=ps, dt=0.01)
    port = CPort(s, workers=2)
    m = o.RK44()
    cache = o.ode_method_cache(m, s)
    u = np.asfortranarray(u0.copy())
    for _ in range(2):
        port.timestep_(m, u, 0.01, cache)
    assert np.allclose(u, st["u"], rtol=1e-12, atol=1e-14)


def test_lmwray3_equals_wray3_tableau(oracle):
    """LMWray3 (step_lmwray3.jl) is the low-storage form of RKMethods.Wray3: on a periodic box the two steppers
    agree to round-off — pins the oracle's low-storage restatement against its (already pinned) ERK loop."""
    o = oracle
    s = fx.setup_periodic(o, (10, 8, 12), D=3, Re=200.0)
    ps = o.psolver_spectral(s)
    u0 = o.random_field(s, kp=2, seed=9, psolver=ps)
    ref = o.solve_unsteady(s, (0.0, 0.02), u0, method=o.Wray3(), psolver=ps, dt=0.01)["u"]
    st = dict(setup=s, psolver=ps, u=u0.copy(order="F"), t=0.0, n=0)
    cache = o.ode_method_cache(o.Wray3(), s)
    for _ in range(2):
        st = o.timestep_lmwray3_(st, 0.01, cache)
    assert np.allclose(st["u"], ref, rtol=1e-11, atol=1e-13)
    assert st["t"] == pytest.approx(0.02) and st["n"] == 2


def _rell2(a, b):
    return float(np.sqrt(np.sum((a - b) ** 2)) / max(np.sqrt(np.sum(b**2)), 1e-300))


def test_numpy_fast_diagonalisation_is_the_oracles_direct_solver(oracle):
    """Pins the helper above to oracle.psolver_direct (sparse LU of laplacian_mat, bordered when singular) on small boxes of the three mid-size
    geometries, consistent and inconsistent right-hand sides."""
    from tests.test_gpu_parity import _numpy_fast_diagonalisation as numpy_fast_diagonalisation

    o = oracle
    lid = (1.0, 0.2, 0.0)
    D_, P_ = o.DirichletBC, o.PeriodicBC
    for x, bcs in (
        ((o.cosine_grid(0.0, 1.0, 8), o.cosine_grid(0.0, 1.0, 6), np.linspace(-0.2, 0.2, 5)), ((D_(), D_()), (D_(), D_(lid)), (P_(), P_()))),
        ((np.linspace(0.0, 2.0, 7), o.tanh_grid(0.0, 1.0, 8, 1.5), np.linspace(0.0, 1.0, 5)), ((P_(), P_()), (D_(), D_()), (P_(), P_()))),
        ((o.tanh_grid(0.0, 1.0, 8, 1.2), o.cosine_grid(0.0, 1.0, 6), o.tanh_grid(0.0, 0.5, 5, 1.1)), ((D_(), D_()), (D_(), D_(lid)), (D_(), D_()))),
    ):
        so = o.make_setup(x, bcs, Re=200.0)
        g = so.grid
        ip = tuple(slice(lo, hi) for lo, hi in g.Ip)
        mine, ref = numpy_fast_diagonalisation(o, so), o.psolver_direct(so)
        for seed in (1, 2):
            f = fx.randn_field(g.N, seed)
            a, b = mine(f.copy(order="F"))[ip], ref(f.copy(order="F"))[ip]
            assert _rell2(a - a.mean(), b - b.mean()) < 1e-10




