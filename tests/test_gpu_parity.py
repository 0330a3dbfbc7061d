"""GPU parity: every HIP entry point (called through the C ABI via the host mirror) against the CPU
oracle on the same seeded inputs, on the reference's own test geometries plus periodic boxes.

Tolerances: single operator <= 1e-12 relative max-norm (reciprocal-multiply instead of
divide, FMA contraction); Poisson solve <= 1e-11 relative L2; multi-step RK4 <= 1e-10 relative L2.
"""
import math

import numpy as np
import pytest

from tests import fixtures as fx

pytestmark = pytest.mark.gpu

OP_TOL = 1e-12
POISSON_TOL = 1e-11
STEP_TOL = 1e-10


@pytest.fixture(scope="module")
def ins():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ins_amd

    return ins_amd


def mirror(ins, so, o):
    """Build the product Setup that corresponds to an oracle setup."""
    cls = {"PeriodicBC": ins.PeriodicBC, "SymmetricBC": ins.SymmetricBC, "PressureBC": ins.PressureBC}

    def conv(b):
        if isinstance(b, o.DirichletBC):
            return ins.DirichletBC(b.u)
        return cls[type(b).__name__]()

    bcs = tuple(tuple(conv(b) for b in side) for side in so.boundary_conditions)
    xin = []
    for a in range(so.grid.D):
        lo = 2 if isinstance(so.boundary_conditions[a][0], o.PressureBC) else 1
        xin.append(so.grid.x[a][lo:-1])
    return ins.Setup(x=xin, boundary_conditions=bcs, Re=so.Re)


def relmax(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def rell2(a, b):
    return float(np.sqrt(np.sum((a - b) ** 2)) / max(np.sqrt(np.sum(b**2)), 1e-300))


def periodic_u0(n):
    return lambda o: fx.setup_periodic(o, n, D=len(n))


GEOMS = {
    "dirichlet2d": fx.setup2d,
    "dirichlet3d": fx.setup3d,
    "mixed3d": fx.setup_mixed,
    "periodic2d": periodic_u0((24, 18)),
    "periodic3d": periodic_u0((20, 12, 70)),  # ragged: x < one wavefront, z > one tile
    "periodic3d_wide": periodic_u0((130, 6, 8)),  # x spans three wavefronts
}


@pytest.mark.parametrize("geom", list(GEOMS))
def test_operators_match_oracle(ins, oracle, geom):
    o = oracle
    so = GEOMS[geom](o)
    sp = mirror(ins, so, o)
    g = so.grid
    D = g.D
    u_h = o.apply_bc_u(fx.randn_field(g.N + (D,), 1), 0.0, so)
    p_h = o.apply_bc_p(fx.randn_field(g.N, 2), 0.0, so)
    F0_h = fx.randn_field(g.N + (D,), 3)
    u, p = ins.from_numpy(sp, u_h), ins.from_numpy(sp, p_h)

    # ghost fills on raw random data (every BC type of the geometry)
    raw_u, raw_p = fx.randn_field(g.N + (D,), 4), fx.randn_field(g.N, 5)
    assert np.array_equal(ins.to_numpy(ins.apply_bc_u_(ins.from_numpy(sp, raw_u), 0.0, sp)), o.apply_bc_u(raw_u, 0.0, so))
    assert np.array_equal(ins.to_numpy(ins.apply_bc_p_(ins.from_numpy(sp, raw_p), 0.0, sp)), o.apply_bc_p(raw_p, 0.0, so))

    assert relmax(ins.to_numpy(ins.divergence(u, sp)), o.divergence(u_h, so)) < OP_TOL
    assert relmax(ins.to_numpy(ins.scalewithvolume(p, sp)), o.scalewithvolume(p_h, so)) < OP_TOL
    assert relmax(ins.to_numpy(ins.pressuregradient(p, sp)), o.pressuregradient(p_h, so)) < OP_TOL
    assert relmax(ins.to_numpy(ins.applypressure(u, p, sp)), o.applypressure_(u_h.copy(order="F"), p_h, so)) < OP_TOL
    assert relmax(ins.to_numpy(ins.laplacian(p, sp)), o.laplacian(p_h, so)) < OP_TOL
    # accumulate-into-F semantics of convection!/diffusion!/convectiondiffusion!
    for name, kw in (("convection_", {}), ("diffusion_", {}), ("diffusion_", {"use_viscosity": False}), ("convectiondiffusion_", {})):
        want = getattr(o, name)(F0_h.copy(order="F"), u_h, so, **kw)
        got = ins.to_numpy(getattr(ins, name)(ins.from_numpy(sp, F0_h), u, sp, **kw))
        assert relmax(got, want) < OP_TOL, name
    # momentum! overwrites F (fill!(F,0) fused): start from garbage
    got = ins.to_numpy(ins.momentum_(ins.from_numpy(sp, F0_h), u, None, 0.0, sp))
    assert relmax(got, o.momentum(u_h, None, 0.0, so)) < OP_TOL
    for first in (False, True):
        ke = ins.to_numpy(ins.kinetic_energy(u, sp, interpolate_first=first))
        assert relmax(ke, o.kinetic_energy_(o.scalarfield(so), u_h, so, interpolate_first=first)) < OP_TOL
        assert ins.total_kinetic_energy(u, sp, interpolate_first=first) == pytest.approx(
            o.total_kinetic_energy(u_h, so, interpolate_first=first), rel=1e-12
        )
    ip = tuple(slice(lo, hi) for lo, hi in g.Ip)
    assert ins.max_abs_divergence(u, sp) == pytest.approx(float(np.max(np.abs(o.divergence(u_h, so)[ip]))), rel=1e-12)
    assert ins.get_cfl_timestep_(None, u, sp) == pytest.approx(o.get_cfl_timestep(u_h, so), rel=1e-12)


def test_dirichlet_constant_and_callable_bc(ins, oracle):
    """apply_bc_u! with tuple constants and with a closure bc.u(α, x..., t) incl. the dudt variant
    (boundary_conditions.jl:344-375); lid-driven-cavity style (examples/LidDrivenCavity3D.jl)."""
    o = oracle
    lid = (1.0, 0.2, 0.0)

    def wall(al, x, y, z, t):
        return (al == 0) * np.cos(t) * np.sin(np.pi * z) + 0 * (x + y + z)

    for top, bot in ((o.DirichletBC(lid), ins.DirichletBC(lid)), (o.DirichletBC(wall), ins.DirichletBC(wall))):
        x = (o.cosine_grid(0.0, 1.0, 9), o.cosine_grid(0.0, 1.0, 7), np.linspace(-0.2, 0.2, 6))
        so = o.make_setup(x, ((o.DirichletBC(), o.DirichletBC()), (o.DirichletBC(), top), (o.PeriodicBC(), o.PeriodicBC())), Re=100.0)
        sp = ins.Setup(x=x, boundary_conditions=((ins.DirichletBC(), ins.DirichletBC()), (ins.DirichletBC(), bot),
                                                 (ins.PeriodicBC(), ins.PeriodicBC())), Re=100.0)
        raw = fx.randn_field(so.grid.N + (3,), 9)
        for dudt in (False, True):
            want = o.apply_bc_u(raw, 0.3, so, dudt=dudt)
            got = ins.to_numpy(ins.apply_bc_u_(ins.from_numpy(sp, raw), 0.3, sp, dudt=dudt))
            assert np.allclose(got, want, rtol=1e-14, atol=1e-14)


@pytest.mark.parametrize("method", ["RK44", "Wray3"])
def test_time_dependent_wall_data_in_the_native_stage_loop(ins, oracle, method, monkeypatch):
    """A moving lid bc.u(α, x..., t) (boundary_conditions.jl:351-357): the stage loop fills ghost volumes at tstart and tstart + c[i] Δt, so the host evaluates
    the closure for those times before the step and the loop runs natively (ins_rk_step_bc_f64) — against the host-driven loop (the reference's own call
    sequence with the closure evaluated between the kernels, INS_HOST_STAGE_LOOP=1) and against the oracle's stage loop."""
    o = oracle

    def lid(al, x, y, z, t):
        return (al == 0) * (1.0 + 0.5 * np.sin(3.0 * t)) * np.sin(np.pi * x) ** 2 + (al == 2) * 0.2 * np.cos(2.0 * t) + 0 * (x + y + z)

    x = (o.cosine_grid(0.0, 1.0, 12), o.cosine_grid(0.0, 1.0, 10), np.linspace(-0.2, 0.2, 9))
    so = o.make_setup(x, ((o.DirichletBC(), o.DirichletBC()), (o.DirichletBC(), o.DirichletBC(lid)), (o.PeriodicBC(), o.PeriodicBC())), Re=100.0)
    sp = ins.Setup(x=x, boundary_conditions=((ins.DirichletBC(), ins.DirichletBC()), (ins.DirichletBC(), ins.DirichletBC(lid)),
                                             (ins.PeriodicBC(), ins.PeriodicBC())), Re=100.0)
    pso, psp = o.psolver_direct(so), ins.psolver_direct(sp)
    u0 = o.apply_bc_u(np.zeros(so.grid.N + (3,), order="F"), 0.0, so)
    mo, mp_ = getattr(o, method)(), getattr(ins.RKMethods, method)()
    want = o.solve_unsteady(so, (0.0, 0.06), u0, method=mo, psolver=pso, dt=0.02)["u"]

    def run():
        (u, _, t), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 0.06), ustart=ins.from_numpy(sp, u0), method=mp_, psolver=psp, Δt=0.02)
        assert t == pytest.approx(0.06)
        return ins.to_numpy(u)

    native = run()
    monkeypatch.setenv("INS_HOST_STAGE_LOOP", "1")
    host = run()
    assert np.abs(host).max() > 0.1
    assert rell2(native, host) < 1e-12
    assert rell2(native, want) < 1e-8  # (the lid's normal component makes the bordered system inconsistent: both sides solve it in the least-squares sense)


# ------------------------------------------------------------------ test/psolvers.jl:1-32 on the GPU
def test_pressure_solvers_known_answer(ins, oracle):
    o = oracle
    so = fx.setup_psolver(o, 32)
    sp = mirror(ins, so, o)
    g = so.grid
    p_exact = np.asfortranarray(0.25 * (np.cos(2 * g.xp[0].reshape(-1, 1)) + np.cos(2 * g.xp[1].reshape(1, -1))))
    o.apply_bc_p_(p_exact, 0.0, so)
    pe = ins.from_numpy(sp, p_exact)
    lap = ins.laplacian(pe, sp)
    ip = tuple(slice(lo, hi) for lo, hi in g.Ip)
    for mk in (ins.psolver_direct, ins.psolver_cg, ins.psolver_spectral):  # test/psolvers.jl:21-30
        got = ins.to_numpy(ins.apply_bc_p(ins.poisson(mk(sp), lap), 0.0, sp))
        assert np.allclose(got[ip], p_exact[ip], rtol=math.sqrt(o.EPS), atol=1e-9)


def _channel(o):
    x = (np.linspace(0.0, 2.0, 9), o.tanh_grid(0.0, 1.0, 6, 1.5), o.stretched_grid(0.0, 1.0, 5, 1.2))
    bcs = ((o.PeriodicBC(), o.PeriodicBC()), (o.DirichletBC(), o.DirichletBC()), (o.SymmetricBC(), o.PressureBC()))
    return o.make_setup(x, bcs, Re=100.0)


def _zperiodic(o, kind):
    """Walls / open sides in x and y, a periodic uniform power-of-two z: the direct solver's fused z pass (Fourier modes instead of Vz)."""
    x = (o.cosine_grid(0.0, 1.0, 12), o.tanh_grid(0.0, 1.0, 10, 1.3), np.linspace(-0.2, 0.2, 33))
    D, P, S, W = o.DirichletBC, o.PeriodicBC, o.SymmetricBC, o.PressureBC
    if kind in ("zwall", "zopen"):  # periodic x and y, walls (or an open top) in z: the orientation of the reference's TurbulentChannel.jl
        x = (np.linspace(0.0, 2 * np.pi, 33), np.linspace(0.0, 1.0, 17), o.tanh_grid(0.0, 1.0, 11, 1.4))
        return o.make_setup(x, ((P(), P()), (P(), P()), (D(), D() if kind == "zwall" else W())), Re=100.0)
    if kind in ("dctwalls", "dctsym", "dctopen"):  # a UNIFORM power-of-two z between walls: the fused z pass in its cosine (DCT) form
        x = (o.cosine_grid(0.0, 1.0, 12), o.tanh_grid(0.0, 1.0, 10, 1.3), np.linspace(0.0, 0.5, 33))
        zb = (S(), S()) if kind == "dctsym" else (D(), D((0.1, 0.0, 0.0)))
        xb = (D(), W()) if kind == "dctopen" else (D(), D())  # dctopen: an open side makes the system regular
        return o.make_setup(x, (xb, (D(), D((1.0, 0.0, 0.2))), zb), Re=100.0)
    bcs = {"cavity": ((D(), D()), (D(), D((1.0, 0.0, 0.2))), (P(), P())),
           "open": ((D(), W()), (S(), S()), (P(), P())),
           "xyper": ((P(), P()), (D(), D()), (P(), P())),
           "channel": ((P(), P()), (D(), W()), (P(), P()))}[kind]
    if kind == "xyper":
        x = (np.linspace(0.0, 1.0, 17), x[1], np.linspace(0.0, 2 * np.pi, 65))
    if kind == "channel":  # open top: a regular (non-singular) system with Fourier x and z
        x = (np.linspace(0.0, 2 * np.pi, 33), o.tanh_grid(0.0, 2.0, 12, 1.5), np.linspace(0.0, np.pi, 33))
    return o.make_setup(x, bcs, Re=100.0)


@pytest.mark.parametrize("geom", ["dirichlet2d", "dirichlet3d", "mixed3d", "periodic2d", "periodic3d", "channel3d", "z:cavity", "z:open", "z:xyper", "z:channel", "z:zwall", "z:zopen",
                                  "z:dctwalls", "z:dctsym", "z:dctopen"])
@pytest.mark.parametrize("consistent", [True, False])
def test_direct_matches_oracle_direct(ins, oracle, geom, consistent):
    """psolver_direct (fast diagonalisation on rocBLAS) against the oracle's sparse-LU factorisation of laplacian_mat,
    including right-hand sides outside the range of a singular L (the bordered system, pressure.jl:133-140)."""
    o = oracle
    if geom == "periodic3d":  # the oracle's sparse LU is what costs here (27 s on the 20 x 12 x 70 box of GEOMS): the same ragged shape class, fewer volumes
        so = fx.setup_periodic(o, (20, 12, 34), D=3)
    else:
        so = _channel(o) if geom == "channel3d" else (_zperiodic(o, geom[2:]) if geom.startswith("z:") else GEOMS[geom](o))
    sp = mirror(ins, so, o)
    g = so.grid
    if consistent:
        u_h = o.apply_bc_u(fx.randn_field(g.N + (g.D,), 8), 0.0, so)
        f = o.scalewithvolume(o.divergence(u_h, so), so)
    else:
        f = fx.randn_field(g.N, 9)
    want = o.poisson(o.psolver_direct(so), f)
    solver = ins.psolver_direct(sp)
    assert solver.kind == "direct"
    if geom.startswith("z:dct"):  # the cosine form of the fused z pass really runs
        from ins_amd import _lib

        assert _lib.load().ins_dbg_fdm_modes(solver.handle) & 8
    got = ins.to_numpy(ins.poisson(solver, ins.from_numpy(sp, f)))
    ip = tuple(slice(lo, hi) for lo, hi in g.Ip)
    assert rell2(got[ip], want[ip]) < POISSON_TOL
    mask = np.ones(g.N, bool)
    mask[ip] = False
    assert np.array_equal(got[mask], f[mask])  # only view(p, Ip) is written (pressure.jl:150)
    # and it really solves L p = f - mean(f)·[singular] with the device Laplacian
    pb = ins.apply_bc_p(ins.from_numpy(sp, got), 0.0, sp)
    res = ins.to_numpy(ins.laplacian(pb, sp))[ip] - f[ip]
    singular = not any(isinstance(b, o.PressureBC) for side in so.boundary_conditions for b in side)
    if singular:
        res += f[ip].mean()
    assert np.abs(res).max() < 1e-10 * max(np.abs(f[ip]).max(), 1.0)


@pytest.mark.parametrize("n", [(16, 16), (12, 20, 8), (64, 32, 16)])
def test_spectral_poisson_matches_oracle(ins, oracle, n):
    o = oracle
    so = fx.setup_periodic(o, n, D=len(n))
    sp = mirror(ins, so, o)
    f = fx.randn_field(so.grid.N, 6)
    ip = tuple(slice(lo, hi) for lo, hi in so.grid.Ip)
    f[ip] -= f[ip].mean()
    want = o.poisson(o.psolver_spectral(so), f)
    got = ins.to_numpy(ins.poisson(ins.psolver_spectral(sp), ins.from_numpy(sp, f)))
    assert rell2(got[ip], want[ip]) < POISSON_TOL
    # ghost entries are left untouched by the solver (pressure.jl:347 writes view(p, Ip) only)
    mask = np.ones(so.grid.N, bool)
    mask[ip] = False
    assert np.array_equal(got[mask], f[mask])


def test_spectral_rejects_nonperiodic_and_odd(ins, oracle):
    o = oracle
    sp = mirror(ins, fx.setup2d(o), o)
    with pytest.raises(ins.INSHipError, match="periodic"):
        ins.psolver_spectral(sp)
    sp = ins.Setup(x=(np.linspace(0, 1, 8), np.linspace(0, 1, 9)))
    with pytest.raises(ins.INSHipError, match="even"):
        ins.psolver_spectral(sp)
    sp = ins.Setup(x=(o.cosine_grid(0.0, 1.0, 8), np.linspace(0, 1, 9)))
    with pytest.raises(ins.INSHipError, match="uniform"):
        ins.psolver_spectral(sp)


@pytest.mark.parametrize("geom", ["dirichlet2d", "dirichlet3d", "mixed3d", "periodic3d"])
def test_cg_matches_oracle_cg(ins, oracle, geom):
    o = oracle
    so = GEOMS[geom](o)
    sp = mirror(ins, so, o)
    g = so.grid
    u_h = o.apply_bc_u(fx.randn_field(g.N + (g.D,), 8), 0.0, so)
    f = o.scalewithvolume(o.divergence(u_h, so), so)
    info = {}
    want = o.poisson(o.psolver_cg(so, info=info), f)
    solver = ins.psolver_cg(sp)
    got = ins.to_numpy(ins.poisson(solver, ins.from_numpy(sp, f)))
    ip = tuple(slice(lo, hi) for lo, hi in g.Ip)
    it, res = solver.last_info()
    assert abs(it - info["iterations"]) <= 2
    assert rell2(got[ip], want[ip]) < 1e-6  # both stop at reltol sqrt(eps); they agree to that level
    # device-resident scalars (default; the host reads the stopping flag once per batch) against the reference's three host reads per
    # iteration (INS_CG_HOSTSYNC), and batch sizes that do / do not divide the iteration count: same iteration count, same iterate
    from ins_amd import _lib

    for opts in ({"INS_CG_HOSTSYNC": 1}, {"INS_CG_BATCH": 1}, {"INS_CG_BATCH": 7}):
        with _lib.options(**opts):
            s2 = ins.psolver_cg(sp)
            got2 = ins.to_numpy(ins.poisson(s2, ins.from_numpy(sp, f)))
            it2, res2 = s2.last_info()
        assert it2 == it, opts
        if "INS_CG_HOSTSYNC" in opts:  # other summation order of the dots (3-D block partials): rounding, amplified by cond(L) on stretched grids
            assert rell2(got2[ip], got[ip]) < 1e-8, opts
        else:  # same kernels, only the host's look at the flag moves: bitwise the same iterate
            assert np.array_equal(got2, got), opts


def _pressure_left(o, D=2):  # PressureBC on LEFT sides (two ghost layers there) next to Dirichlet / Symmetric / Pressure
    x = (o.cosine_grid(0.0, 1.0, 7), np.linspace(0.0, 1.0, 6), o.tanh_grid(0.0, 0.5, 5, 1.2))[:D]
    bcs = ((o.PressureBC(), o.DirichletBC()), (o.PressureBC(), o.PressureBC()), (o.SymmetricBC(), o.PressureBC()))[:D]
    return o.make_setup(x, bcs, Re=100.0)


def _mixed2d(o):  # Periodic x (Symmetric | Pressure): every ghost rule of apply_bc_p! in two dimensions
    x = (np.linspace(0.0, 2.0, 11), o.tanh_grid(0.0, 1.0, 9, 1.4))
    return o.make_setup(x, ((o.PeriodicBC(), o.PeriodicBC()), (o.SymmetricBC(), o.PressureBC())), Re=200.0)


@pytest.mark.parametrize("geom", ["periodic2d", "periodic3d", "dirichlet2d", "dirichlet3d", "mixed2d", "mixed3d", "channel3d", "pleft2d", "pleft3d"])
def test_project_matches_oracle(ins, oracle, geom):
    o = oracle
    special = {"mixed2d": _mixed2d, "channel3d": _channel, "pleft2d": lambda o: _pressure_left(o, 2), "pleft3d": lambda o: _pressure_left(o, 3)}
    so = special[geom](o) if geom in special else GEOMS[geom](o)
    sp = mirror(ins, so, o)
    g = so.grid
    u_h = o.apply_bc_u(fx.randn_field(g.N + (g.D,), 12), 0.0, so)
    periodic = geom.startswith("periodic")
    pso = o.psolver_spectral(so) if periodic else o.psolver_direct(so)
    psp = ins.psolver_spectral(sp) if periodic else ins.psolver_direct(sp)
    want_u = o.project_(u_h.copy(order="F"), so, pso, o.scalarfield(so))
    u, p = ins.from_numpy(sp, u_h), ins.scalarfield(sp)
    ins.project_(u, sp, psp, p)
    tol = POISSON_TOL
    assert rell2(ins.to_numpy(u), want_u) < tol
    ins.apply_bc_u_(u, 0.0, sp)
    if periodic:
        assert ins.max_abs_divergence(u, sp) < 1e-9
    # out-of-place twin == in-place (pressure.jl:52-66 vs 69-82)
    v = ins.project(ins.from_numpy(sp, u_h), sp, psp)
    assert rell2(ins.to_numpy(v), want_u) < tol
    if periodic:  # p is left ghost-filled, like apply_bc_p!(p) in project!
        want_p = o.scalarfield(so)
        o.project_(u_h.copy(order="F"), so, pso, want_p)
        assert rell2(ins.to_numpy(p), want_p) < POISSON_TOL


# ------------------------------------------------------------------ RK stepping
@pytest.mark.parametrize("method", ["RK44", "Wray3", "SSP33", "FE11"])
def test_rk_step_matches_oracle_periodic3d(ins, oracle, method):
    o = oracle
    so = fx.setup_periodic(o, (16, 12, 20), D=3, Re=500.0)
    sp = mirror(ins, so, o)
    pso, psp = o.psolver_spectral(so), ins.psolver_spectral(sp)
    u0 = o.random_field(so, kp=3, seed=2, psolver=pso)
    mo, mp = getattr(o, method)(), getattr(ins.RKMethods, method)()
    st = o.solve_unsteady(so, (0.0, 0.03), u0, method=mo, psolver=pso, dt=0.01)
    (u, _, t), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 0.03), ustart=ins.from_numpy(sp, u0), method=mp, psolver=psp, Δt=0.01)
    assert t == pytest.approx(0.03)
    assert rell2(ins.to_numpy(u), st["u"]) < STEP_TOL
    assert ins.max_abs_divergence(u, sp) < 1e-11


def test_rk44_tgv3d_64_ten_steps(ins, oracle):
    """10 RK4 steps of TGV3D 64³ <= 1e-10 relative L2; max|div u|·Δx <= 1e-12."""
    o = oracle
    so = fx.setup_periodic(o, 64, D=3, Re=1000.0)
    sp = mirror(ins, so, o)
    pso, psp = o.psolver_spectral(so), ins.psolver_spectral(sp)
    u0 = o.velocityfield(so, o.tgv3d_ufunc, 0.0, psolver=pso)
    up = ins.velocityfield(sp, o.tgv3d_ufunc, 0.0, psolver=psp)
    assert rell2(ins.to_numpy(up), u0) < POISSON_TOL
    st = o.solve_unsteady(so, (0.0, 0.01), u0, psolver=pso, dt=1e-3)
    (u, _, t), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 0.01), ustart=up, psolver=psp, Δt=1e-3)
    assert rell2(ins.to_numpy(u), st["u"]) < STEP_TOL
    assert ins.max_abs_divergence(u, sp) / 64 < 1e-12
    assert ins.total_kinetic_energy(u, sp) == pytest.approx(o.total_kinetic_energy(st["u"], so), rel=1e-11)


def test_rk44_dirichlet_cavity_matches_oracle(ins, oracle):
    """Config-5-shaped problem at test size: stretched grid, lid-driven Dirichlet walls, periodic z, default (direct) solver."""
    o = oracle
    lid = (1.0, 0.2, 0.0)
    x = (o.cosine_grid(0.0, 1.0, 12), o.cosine_grid(0.0, 1.0, 10), np.linspace(-0.2, 0.2, 9))
    so = o.make_setup(x, ((o.DirichletBC(), o.DirichletBC()), (o.DirichletBC(), o.DirichletBC(lid)), (o.PeriodicBC(), o.PeriodicBC())), Re=100.0)
    sp = ins.Setup(x=x, boundary_conditions=((ins.DirichletBC(), ins.DirichletBC()), (ins.DirichletBC(), ins.DirichletBC(lid)),
                                             (ins.PeriodicBC(), ins.PeriodicBC())), Re=100.0)
    pso, psp = o.psolver_direct(so), ins.default_psolver(sp)
    assert psp.kind == "direct"
    u0 = o.apply_bc_u_(o.vectorfield(so), 0.0, so)
    st = o.solve_unsteady(so, (0.0, 0.02), u0, psolver=pso, dt=0.005)
    (u, _, t), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 0.02), ustart=ins.from_numpy(sp, u0), psolver=psp, Δt=0.005)
    assert rell2(ins.to_numpy(u), st["u"]) < STEP_TOL
    # the bordered CG variant solves the same system (slower): agrees to its tolerance
    (u2, _, _), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 0.02), ustart=ins.from_numpy(sp, u0),
                                       psolver=ins.psolver_cg(sp, reltol=1e-12, bordered=True), Δt=0.005)
    assert rell2(ins.to_numpy(u2), st["u"]) < 1e-8


def test_timestep_inplace_equals_outofplace(ins, oracle):
    """test/timesteppers.jl:16-42"""
    o = oracle
    so = fx.setup_periodic(o, 16, D=2)
    sp = mirror(ins, so, o)
    psp = ins.psolver_spectral(sp)
    u = ins.random_field(sp, kp=4, psolver=psp, seed=3)
    m = ins.RKMethods.RK44()
    s_out = ins.timestep(m, ins.create_stepper(m, setup=sp, psolver=psp, u=ins.copyfield(u), t=0.0), 0.1)
    cache = ins.ode_method_cache(m, sp, psp)
    s_in = ins.timestep_(m, ins.create_stepper(m, setup=sp, psolver=psp, u=ins.copyfield(u), t=0.0), 0.1, cache=cache)
    assert np.allclose(ins.to_numpy(s_in.u), ins.to_numpy(s_out.u), rtol=1e-13, atol=1e-14)
    assert s_in.n == 1 and s_in.t == pytest.approx(0.1)


def test_random_field_is_solenoidal_with_right_spectrum_peak(ins):
    sp = ins.Setup(x=(np.linspace(0, 1, 33),) * 3, Re=4000.0)
    u = ins.random_field(sp, kp=4, seed=0)
    assert ins.max_abs_divergence(u, sp) < 1e-10
    e = ins.total_kinetic_energy(u, sp)
    assert 0 < e < 10
    u2 = ins.random_field(sp, kp=4, seed=0)
    assert np.array_equal(ins.to_numpy(u), ins.to_numpy(u2))  # seeded => reproducible


# ------------------------------------------------------------------ examples/TaylorGreenVortex2D.jl on the GPU
def test_tgv2d_convergence_on_gpu(ins, oracle):
    o = oracle
    Re, tend = 2000.0, 0.5
    errs = []
    for n in (8, 16, 32):
        x = (np.linspace(0, 2 * np.pi, n + 1),) * 2
        sp = ins.Setup(x=x, Re=Re)
        ps = ins.psolver_spectral(sp)
        sol = o.tgv2d_ufunc(Re)
        u0 = ins.velocityfield(sp, sol(0.0), 0.0, psolver=ps)
        ut = ins.to_numpy(ins.velocityfield(sp, sol(tend), tend, psolver=ps, doproject=False))
        (u, _, _), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, tend), ustart=u0, psolver=ps, Δt=0.01)
        ip = tuple(slice(lo, hi) for lo, hi in sp.grid.Ip)
        un = ins.to_numpy(u)
        errs.append(math.sqrt(np.sum((un[ip] - ut[ip]) ** 2)) / math.sqrt(np.sum(ut[ip] ** 2)))
        del ps, sp, u, u0  # one solver's rocFFT plans alive at a time (csrc/ins_fftcheck.hip)
    assert errs[0] == pytest.approx(2.518e-5, rel=2e-3) and errs[1] == pytest.approx(6.393e-6, rel=2e-3)
    assert errs[2] == pytest.approx(1.604e-6, rel=2e-3)


# ------------------------------------------------------------------ full-size properties (BASELINE sizes)
def test_full_size_256_properties(ins):
    """TGV3D 256³ (BASELINE config 2): size-independent properties — projection idempotence,
    divergence-free stage velocities, energy decay, momentum linearity in viscosity."""
    n = 256
    sp = ins.Setup(x=(np.linspace(0, 1, n + 1),) * 3, Re=1000.0)
    ps = ins.psolver_spectral(sp)

    def U(al, x, y, z):
        if al == 0:
            return np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y) * np.sin(2 * np.pi * z) / 2
        if al == 1:
            return -np.cos(2 * np.pi * x) * np.sin(2 * np.pi * y) * np.sin(2 * np.pi * z) / 2
        return 0 * (x + y + z)

    u = ins.velocityfield(sp, U, 0.0, psolver=ps)
    assert ins.max_abs_divergence(u, sp) / n < 1e-12
    e0 = ins.total_kinetic_energy(u, sp)
    assert e0 == pytest.approx(1 / 32, rel=2e-3)  # ∫ (u²+v²)/2 over the unit box for this field
    v = ins.copyfield(u)
    ins.project_(v, sp, ps, ins.scalarfield(sp))
    assert float((v - u).abs().max()) < 1e-13  # P² = P
    (w, _, _), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 2e-3), ustart=u, psolver=ps, Δt=1e-3)
    assert ins.max_abs_divergence(w, sp) / n < 1e-12
    assert ins.total_kinetic_energy(w, sp) < e0
    # momentum is affine in 1/Re: F(Re1) - F(Re2) = (1/Re1 - 1/Re2) * diffusion(u; use_viscosity=false)
    sp2 = ins.Setup(x=(np.linspace(0, 1, n + 1),) * 3, Re=10.0)
    F1, F2 = ins.momentum(u, None, 0.0, sp), ins.momentum(u, None, 0.0, sp2)
    Dm = ins.diffusion(u, sp, use_viscosity=False)
    lhs, rhs = (F2 - F1), (1 / 10.0 - 1 / 1000.0) * Dm
    assert float((lhs - rhs).abs().max()) < 1e-9 * float(rhs.abs().max())


@pytest.mark.parametrize("nz", [16, 32, 64, 128, 192, 256, 384, 512, 1024])
def test_fused_z_pass_matches_oracle(ins, oracle, nz):
    """The custom z kernel (DIF FFT · symbol · DIT inverse FFT, csrc/ins_zsolve.hip) for every supported nz,
    including the odd-log2 sizes that take the extra radix-2 stage and 192 / 384 (3 x 8 x 8, 6 x 8 x 8 in the three-pass
    kernel); ragged line count (kxn*ny not a tile multiple)."""
    o = oracle
    n = (10, 6, nz)
    so = fx.setup_periodic(o, n, D=3)
    sp = mirror(ins, so, o)
    f = fx.randn_field(so.grid.N, 21)
    ip = tuple(slice(lo, hi) for lo, hi in so.grid.Ip)
    f[ip] -= f[ip].mean()
    want = o.poisson(o.psolver_spectral(so), f)
    got = ins.to_numpy(ins.poisson(ins.psolver_spectral(sp), ins.from_numpy(sp, f)))
    assert rell2(got[ip], want[ip]) < POISSON_TOL


@pytest.mark.parametrize("geom", ["periodic3d", "dirichlet3d", "periodic2d"])
def test_lmwray3_and_right_hand_side_match_oracle(ins, oracle, geom):
    """§8f: LMWray3 (step_lmwray3.jl:4-107) and right_hand_side! (sciml.jl:35-47) — same kernels, re-orchestrated."""
    o = oracle
    so = GEOMS[geom](o)
    sp = mirror(ins, so, o)
    g = so.grid
    periodic = geom.startswith("periodic")
    pso = o.psolver_spectral(so) if periodic else o.psolver_direct(so)
    psp = ins.psolver_spectral(sp) if periodic else ins.default_psolver(sp)
    u0 = o.project(o.apply_bc_u(0.1 * fx.randn_field(g.N + (g.D,), 31), 0.0, so), so, pso)
    o.apply_bc_u_(u0, 0.0, so)
    tol = STEP_TOL if periodic else 1e-7
    # right_hand_side!
    want = o.right_hand_side(u0, so, pso, 0.0)
    dudt = ins.vectorfield(sp)
    u_dev = ins.from_numpy(sp, u0)
    ins.right_hand_side_(dudt, u_dev, (sp, psp), 0.0)
    assert rell2(ins.to_numpy(dudt), want) < tol
    assert np.array_equal(ins.to_numpy(u_dev), u0)  # "be careful to not touch u in this function" (sciml.jl:41)
    assert rell2(ins.to_numpy(ins.create_right_hand_side(sp, psp)(u_dev, None, 0.0)), want) < tol
    # LMWray3
    st = dict(setup=so, psolver=pso, u=u0.copy(order="F"), t=0.0, n=0)
    cache = o.ode_method_cache(o.Wray3(), so)
    for _ in range(2):
        st = o.timestep_lmwray3_(st, 0.005, cache)
    m = ins.LMWray3()
    (u, _, t), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 0.01), ustart=u_dev, method=m, psolver=psp, Δt=0.005)
    assert t == pytest.approx(0.01)
    assert rell2(ins.to_numpy(u), st["u"]) < tol


@pytest.mark.parametrize("geom", ["periodic3d", "periodic3d_wide", "dirichlet3d", "periodic2d"])
def test_lmwray3_native_loop_equals_host_driven_loop(ins, oracle, geom, monkeypatch):
    """LMWray3 runs inside the native stage loop as the explicit RK method its own tableau comment describes (step_lmwray3.jl:65-76; time_steppers.py
    `_lmwray3_as_erk`): single steps and the chained `timesteps_` against the host-driven loop that issues the reference's own axpy sequence
    (INS_HOST_STAGE_LOOP=1) and against the oracle's restatement of that sequence."""
    o = oracle
    so = GEOMS[geom](o)
    sp = mirror(ins, so, o)
    g = so.grid
    periodic = geom.startswith("periodic")
    pso = o.psolver_spectral(so) if periodic else o.psolver_direct(so)
    psp = ins.psolver_spectral(sp) if periodic else ins.default_psolver(sp)
    u0 = o.project(o.apply_bc_u(0.1 * fx.randn_field(g.N + (g.D,), 33), 0.0, so), so, pso)
    o.apply_bc_u_(u0, 0.0, so)
    tol = STEP_TOL if periodic else 1e-7
    st = dict(setup=so, psolver=pso, u=u0.copy(order="F"), t=0.0, n=0)
    oc = o.ode_method_cache(o.Wray3(), so)
    for _ in range(3):
        st = o.timestep_lmwray3_(st, 0.004, oc)
    m = ins.LMWray3()

    def run(chained):
        cache = ins.ode_method_cache(m, sp, psp)
        s = ins.create_stepper(m, setup=sp, psolver=psp, u=ins.from_numpy(sp, u0), t=0.0)
        if chained:
            s = ins.timesteps_(m, s, 0.004, 3, cache=cache)
        else:
            for _ in range(3):
                s = ins.timestep_(m, s, 0.004, cache=cache)
        assert s.t == pytest.approx(0.012) and s.n == 3
        return ins.to_numpy(s.u), cache

    native, cache = run(False)
    assert cache._erk is not None and cache._host is None  # the native loop ran; the low-storage registers were never allocated
    chained, _ = run(True)
    monkeypatch.setenv("INS_HOST_STAGE_LOOP", "1")
    host, hcache = run(False)
    assert hcache._erk is None and hcache._host
    assert rell2(native, host) < 1e-12 and rell2(chained, host) < 1e-12
    assert rell2(native, st["u"]) < tol


@pytest.mark.parametrize("n", [(16, 16, 16), (32, 16, 64), (128, 32, 16), (64, 128, 32), (256, 16, 16), (16, 256, 32),
                               (512, 16, 16), (16, 512, 16), (1024, 16, 16), (16, 1024, 16),
                               (192, 16, 16), (16, 192, 32), (16, 16, 192), (384, 16, 16), (16, 384, 16), (32, 16, 384), (192, 384, 192),
                               (320, 16, 16), (16, 320, 32), (16, 16, 320), (640, 16, 16), (16, 640, 16), (32, 16, 640), (320, 192, 320),
                               (96, 16, 16), (16, 96, 32), (16, 32, 96), (160, 16, 16), (32, 160, 16), (16, 16, 160), (160, 96, 160), (32, 16, 256), (16, 256, 256)])
def test_own_fft_passes_match_oracle(ins, oracle, n):
    """All-own-kernel spectral solve (csrc/ins_fft.hip: paired-row real x transform, digit-reversed y pass, fused z pass) —
    every supported length incl. the odd-log2 ones, 96 / 192 / 384 (a radix-3 stage in front of the power-of-two stages) and 160 / 320 / 640 (a radix-5
    stage in front), in each direction; both the generic psolver(p) entry and the fused projection (right-hand side formed inside the x pass)."""
    import ctypes
    from ins_amd import _lib

    o = oracle
    so = fx.setup_periodic(o, n, D=3)
    sp = mirror(ins, so, o)
    pso, psp = o.psolver_spectral(so), ins.psolver_spectral(sp)
    engine = ctypes.c_int32(-2)
    _lib.call("ins_poisson_fft_engine", psp.handle, ctypes.byref(engine))
    assert engine.value == 1  # no rocFFT plan in this solver
    f = fx.randn_field(so.grid.N, 23)
    ip = tuple(slice(lo, hi) for lo, hi in so.grid.Ip)
    f[ip] -= f[ip].mean()
    want = o.poisson(pso, f)
    got = ins.to_numpy(ins.poisson(psp, ins.from_numpy(sp, f)))
    assert rell2(got[ip], want[ip]) < POISSON_TOL
    u_h = o.apply_bc_u(fx.randn_field(so.grid.N + (3,), 24), 0.0, so)
    want_u = o.project_(u_h.copy(order="F"), so, pso, o.scalarfield(so))
    u = ins.from_numpy(sp, u_h)
    ins.project_(u, sp, psp, ins.scalarfield(sp))
    assert rell2(ins.to_numpy(u), want_u) < POISSON_TOL


@pytest.mark.parametrize("n,P", [((256, 128, 64), 0), ((32, 128, 64), 2), ((32, 128, 64), 8), ((64, 256, 64), 4), ((16, 256, 128), 16), ((32, 256, 128), 4),
                                 ((16, 512, 64), 2), ((32, 512, 128), 8), ((128, 128, 128), 0), ((64, 256, 256), 16)])
def test_four_pass_solve_matches_oracle(ins, oracle, n, P):
    """The z direction of the spectral solve as periodic tridiagonal systems riding on the two y passes (csrc/ins_fft.hip k_yz_fwd / k_yz_iface / k_yz_bwd:
    four passes per solve instead of five): the same linear system as the reference's division by âx + ây + âz (pressure.jl:326-341), here against the
    oracle's FFT solve — every built y length (128, 256, 512), partition counts 2 .. 16 (forced: INS_YZ_PARTITIONS; 0: the library's choice), anisotropic
    spacings (the decaying root of a line ranges from ~1 to ~1e-3), both the psolver(p) entry and the fused projection; and against the five-pass route."""
    import ctypes
    from ins_amd import _lib

    o = oracle
    L = (1.0, 2.0, 0.25) if n[0] != n[2] else (1.0, 1.0, 1.0)
    x = tuple(np.linspace(0.0, L[a], n[a] + 1) for a in range(3))
    so = o.make_setup(x, Re=1000.0)
    sp = ins.Setup(x=x, Re=1000.0)
    pso = o.psolver_spectral(so)
    with _lib.options(INS_YZ_FUSED=1, INS_YZ_PARTITIONS=P):
        psp = ins.psolver_spectral(sp)
    ps5 = ins.psolver_spectral(sp)  # the default: five passes
    parts = ctypes.c_int32(-1)
    _lib.call("ins_poisson_yz_partitions", psp.handle, ctypes.byref(parts))
    assert parts.value == (P if P else parts.value) and parts.value >= 2
    _lib.call("ins_poisson_yz_partitions", ps5.handle, ctypes.byref(parts))
    assert parts.value == 0
    f = fx.randn_field(so.grid.N, 23)
    ip = tuple(slice(lo, hi) for lo, hi in so.grid.Ip)
    f[ip] -= f[ip].mean()
    want = o.poisson(pso, f)
    got = ins.to_numpy(ins.poisson(psp, ins.from_numpy(sp, f)))
    got5 = ins.to_numpy(ins.poisson(ps5, ins.from_numpy(sp, f)))
    assert rell2(got[ip], want[ip]) < POISSON_TOL
    assert rell2(got[ip], got5[ip]) < POISSON_TOL
    assert abs(got[ip].mean()) < 1e-12 * np.abs(got[ip]).max()  # the reference's gauge: zero mean (pressure.jl:336-341)
    u_h = o.apply_bc_u(fx.randn_field(so.grid.N + (3,), 24), 0.0, so)
    want_u = o.project_(u_h.copy(order="F"), so, pso, o.scalarfield(so))
    u = ins.from_numpy(sp, u_h)
    ins.project_(u, sp, psp, ins.scalarfield(sp))
    assert rell2(ins.to_numpy(u), want_u) < POISSON_TOL
    ins.apply_bc_u_(u, 0.0, sp)
    assert ins.max_abs_divergence(u, sp) * min(L[a] / n[a] for a in range(3)) < 1e-11


def test_full_size_512_decaying_turbulence_properties(ins):
    """BASELINE config 3 (DecayingTurbulence3D 512^3, Re = 4000, kp = 10, Δt = 2.5e-4) at full size through
    size-independent properties: solenoidal seeded initial field, divergence-free after stepping (max|div u|·Δx
    <= 1e-12), kinetic energy decays, and two independently seeded runs of the same seed agree bitwise."""
    import torch

    n = 512
    sp = ins.Setup(x=(np.linspace(0, 1, n + 1),) * 3, Re=4000.0)
    ps = ins.psolver_spectral(sp)
    u = ins.random_field(sp, kp=10, A=1.0, seed=0, psolver=ps)
    assert ins.max_abs_divergence(u, sp) / n < 1e-12
    e0 = ins.total_kinetic_energy(u, sp)
    chk0 = float(u.sum())
    (w, _, t), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 5e-4), ustart=u, psolver=ps, Δt=2.5e-4)
    assert t == pytest.approx(5e-4)
    assert ins.max_abs_divergence(w, sp) / n < 1e-12
    e1 = ins.total_kinetic_energy(w, sp)
    assert 0 < e1 < e0
    (w2, _, _), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 5e-4), ustart=u, psolver=ps, Δt=2.5e-4)
    assert torch.equal(w, w2) and float(u.sum()) == chk0  # deterministic, and ustart untouched (docopy)


# ------------------------------------------------------------------ K1, 64 outputs per wavefront (csrc/ins_flux64.hip)
def exact_box(o, n, Re=500.0):
    """Periodic box whose coordinates are exact binary fractions: the metric records are bitwise constant, which is
    what selects the constant-record kernels."""
    return o.make_setup(tuple(np.arange(ni + 1) * 2.0**-6 for ni in n), Re=Re)


# x: one full + one partial wavefront / two full / 3 full + partial / minimum width; y, z: ragged against R and the z-chunk
FLUX64_BOXES = [(96, 9, 7), (128, 16, 12), (200, 8, 5), (66, 10, 4), (130, 13, 9)]


@pytest.mark.parametrize("n", FLUX64_BOXES)
def test_flux64_momentum_matches_oracle(ins, oracle, n):
    from ins_amd import _lib

    o = oracle
    so = exact_box(o, n)
    sp = mirror(ins, so, o)
    assert _lib.load().ins_grid_is_uniform_exact(sp.handle)
    u_h = o.apply_bc_u(fx.randn_field(so.grid.N + (3,), 21), 0.0, so)
    got = ins.to_numpy(ins.momentum_(ins.from_numpy(sp, fx.randn_field(so.grid.N + (3,), 22)), ins.from_numpy(sp, u_h), None, 0.0, sp))
    assert relmax(got, o.momentum(u_h, None, 0.0, so)) < OP_TOL


@pytest.mark.parametrize("n", [(96, 10, 8), (128, 16, 12), (200, 8, 6), (66, 10, 4), (130, 14, 10), (192, 16, 16)])  # even: spectral solver; last: own 3 * 2^m passes
@pytest.mark.parametrize("method", ["RK44", "Wray3"])
def test_flux64_rk_steps_match_oracle(ins, oracle, n, method):
    """Fused stage kernels: first stage (RK epilogue) and the in-register pressure correction of the later stages, on boxes
    whose last wavefront is partial and whose periodic images cross wavefront borders."""
    o = oracle
    so = exact_box(o, n)
    sp = mirror(ins, so, o)
    pso, psp = o.psolver_spectral(so), ins.psolver_spectral(sp)
    u0 = o.random_field(so, kp=2, seed=5, psolver=pso)
    mo, mp_ = getattr(o, method)(), getattr(ins.RKMethods, method)()
    st = o.solve_unsteady(so, (0.0, 0.02), u0, method=mo, psolver=pso, dt=0.01)
    (u, _, t), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 0.02), ustart=ins.from_numpy(sp, u0), method=mp_, psolver=psp, Δt=0.01)
    assert rell2(ins.to_numpy(u), st["u"]) < STEP_TOL
    assert ins.max_abs_divergence(u, sp) < 1e-10
    del psp, sp


@pytest.mark.parametrize("n", [(128, 16, 12), (96, 10, 8), (66, 10, 4), (32, 32, 32)])
def test_chained_steps_equal_single_steps(ins, oracle, n):
    """`timesteps_` (ins_rk_steps_f64): on wide exact boxes the final correction of every step but the last is folded into the next
    step's first stage kernel; the result must be what step-by-step `timestep_` gives (and the oracle); (66, 10, 4) is too thin for the in-kernel
    correction and takes the plain loop, (32, 32, 32) chains on the 62-wide stage kernel (round 3)."""
    o = oracle
    so = exact_box(o, n)
    sp = mirror(ins, so, o)
    pso, psp = o.psolver_spectral(so), ins.psolver_spectral(sp)
    u0 = o.random_field(so, kp=2, seed=9, psolver=pso)
    m = ins.RKMethods.RK44()
    cache = ins.ode_method_cache(m, sp, psp)
    st1 = ins.create_stepper(m, setup=sp, psolver=psp, u=ins.from_numpy(sp, u0), t=0.0)
    for _ in range(5):
        st1 = ins.timestep_(m, st1, 0.01, cache=cache)
    st2 = ins.create_stepper(m, setup=sp, psolver=psp, u=ins.from_numpy(sp, u0), t=0.0)
    st2 = ins.timesteps_(m, st2, 0.01, 5, cache=cache)
    assert st2.n == 5 and st2.t == pytest.approx(0.05)
    a, b = ins.to_numpy(st1.u), ins.to_numpy(st2.u)
    assert rell2(b, a) < 1e-13  # ghosts included
    want = o.solve_unsteady(so, (0.0, 0.05), u0, psolver=pso, dt=0.01)["u"]
    assert rell2(b, want) < STEP_TOL
    assert ins.max_abs_divergence(st2.u, sp) < 1e-10
    # and through solve_unsteady (no processors -> one native call)
    (u3, _, t3), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 0.05), ustart=ins.from_numpy(sp, u0), psolver=psp, Δt=0.01)
    assert rell2(ins.to_numpy(u3), a) < 1e-13
    del psp, sp


@pytest.mark.parametrize("geom", ["mixed3d", "pleft3d", "channel3d", "dirichlet3d"])
def test_rk44_nonperiodic3d_matches_oracle(ins, oracle, geom):
    """The fused non-periodic stage (RK epilogue in the masked stencil kernel, direct solver fused into project!) on every
    boundary-condition mix, against the oracle's reference-ordered stage loop."""
    o = oracle
    special = {"channel3d": _channel, "pleft3d": lambda o: _pressure_left(o, 3)}
    so = special[geom](o) if geom in special else GEOMS[geom](o)
    sp = mirror(ins, so, o)
    pso, psp = o.psolver_direct(so), ins.default_psolver(sp)
    assert psp.kind == "direct"
    u0 = o.apply_bc_u(0.1 * fx.randn_field(so.grid.N + (3,), 31), 0.0, so)
    u0 = o.project_(u0.copy(order="F"), so, pso, o.scalarfield(so))
    u0 = o.apply_bc_u_(u0, 0.0, so)
    st = o.solve_unsteady(so, (0.0, 0.004), u0, psolver=pso, dt=0.001)
    (u, _, t), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 0.004), ustart=ins.from_numpy(sp, u0), psolver=psp, Δt=0.001)
    assert np.isfinite(st["u"]).all()
    assert rell2(ins.to_numpy(u), st["u"]) < STEP_TOL


def test_handles_release_their_device_memory(ins):
    """Create / use / drop every kind of library handle repeatedly: the device's free memory (hipMemGetInfo) must come back, i.e. the
    *_destroy entry points free everything the *_create ones and the lazily built caches allocated."""
    import gc

    import torch

    def cycle(n):
        x = (np.linspace(0.0, 1.0, n[0] + 1), np.linspace(0.0, 1.0, n[1] + 1), np.linspace(0.0, 1.0, n[2] + 1))
        sp = ins.Setup(x=x, Re=1000.0)
        ps = ins.psolver_spectral(sp)
        m = ins.RKMethods.RK44()
        cache = ins.ode_method_cache(m, sp, ps)
        u = ins.random_field(sp, kp=2, psolver=ps, seed=1)
        st = ins.create_stepper(m, setup=sp, psolver=ps, u=u, t=0.0)
        st = ins.timesteps_(m, st, 1e-3, 2, cache=cache)
        ins.observespectrum(dict(u=st.u, temp=None, t=0.0, n=0), setup=sp)
        bc = ((ins.DirichletBC(), ins.DirichletBC()), (ins.DirichletBC(), ins.PressureBC()), (ins.PeriodicBC(), ins.PeriodicBC()))
        s2 = ins.Setup(x=(ins.cosine_grid(0.0, 1.0, n[0]), ins.tanh_grid(0.0, 1.0, n[1]), x[2]), boundary_conditions=bc, Re=100.0)
        for mk in (ins.psolver_direct, ins.psolver_cg):
            p2 = mk(s2)
            c2 = ins.ode_method_cache(m, s2, p2)
            u2 = ins.velocityfield(s2, lambda a, x, y, z: 0 * (x + y + z) + (a == 0), psolver=p2)
            ins.timestep_(m, ins.create_stepper(m, setup=s2, psolver=p2, u=u2, t=0.0), 1e-3, cache=c2)
        torch.cuda.synchronize()

    cycle((64, 32, 32))  # first use: code objects, rocFFT / rocBLAS workspaces
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(6):
        cycle((64, 32, 32))
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 32 * 2**20, f"{(free0 - free1) / 2**20:.1f} MiB not returned after 6 create/destroy cycles"


@pytest.mark.parametrize("n", [(32, 32), (64, 16), (1024, 32), (16, 256), (64, 64), (16, 32), (32, 64)])  # up to 64 x 64: the whole solve is one launch (k_xysolve2d)
def test_own_fft_2d_poisson_matches_oracle_and_rocfft(ins, oracle, n, monkeypatch):
    """2-D power-of-two boxes: own x passes + the fused solve kernel along y, against the oracle and against the rocFFT route
    (INS_DISABLE_OWNFFT, read when the solver is created)."""
    o = oracle
    so = fx.setup_periodic(o, n, D=2, L=2 * np.pi)
    sp = mirror(ins, so, o)
    f = fx.randn_field(so.grid.N, 7)
    want = o.psolver_spectral(so)(f.copy(order="F"))
    got = ins.to_numpy(ins.psolver_spectral(sp)(ins.from_numpy(sp, f)))
    sl = tuple(slice(lo, hi) for lo, hi in so.grid.Ip)
    assert rell2(got[sl], want[sl]) < POISSON_TOL
    monkeypatch.setenv("INS_DISABLE_OWNFFT", "1")
    ref = ins.to_numpy(ins.psolver_spectral(sp)(ins.from_numpy(sp, f)))
    assert rell2(got[sl], ref[sl]) < POISSON_TOL
    monkeypatch.delenv("INS_DISABLE_OWNFFT")
    # and a few RK44 steps through the native loop
    u0 = o.random_field(so, kp=3, seed=5)
    refu = o.solve_unsteady(so, (0.0, 3e-3), u0, dt=1e-3)
    (u, _, _), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 3e-3), ustart=ins.from_numpy(sp, u0), Δt=1e-3)
    assert rell2(ins.to_numpy(u), refu["u"]) < STEP_TOL


@pytest.mark.parametrize("n,L", [((96, 12, 10), 2 * np.pi), ((128, 16, 8), 3.1), ((200, 10, 6), 2 * np.pi)])
def test_nearly_uniform_boxes_take_the_constant_record_kernels(ins, oracle, n, L):
    """Boxes whose spacings are uniform only up to the rounding of the coordinates ([0, 2π]³, 1/96 …) run on the constant-record kernels
    (64-wide stage kernel, in-register correction, stage-velocity basis, chained steps) and still match the table-driven reference arithmetic."""
    import ins_amd

    o = oracle
    x = tuple(np.linspace(0.0, L, ni + 1) for ni in n)
    so = o.make_setup(x, Re=500.0)
    sp = mirror(ins, so, o)
    assert not all(np.ptp(np.diff(xi)) == 0.0 for xi in x)  # not bitwise uniform ...
    assert ins_amd._lib.load().ins_grid_is_uniform_exact(sp.handle) == 1  # ... but classified as constant
    u_h = o.apply_bc_u(fx.randn_field(so.grid.N + (3,), 3), 0.0, so)
    assert relmax(ins.to_numpy(ins.momentum(ins.from_numpy(sp, u_h), None, 0.0, sp)), o.momentum(u_h, None, 0.0, so)) < OP_TOL
    ps_h, ps_d = o.psolver_spectral(so), ins.psolver_spectral(sp)
    u0 = o.random_field(so, kp=2, seed=9, psolver=ps_h)
    ref = o.solve_unsteady(so, (0.0, 4e-3), u0, psolver=ps_h, dt=1e-3)
    (u, _, _), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 4e-3), ustart=ins.from_numpy(sp, u0), psolver=ps_d, Δt=1e-3)  # 4 chained steps
    assert rell2(ins.to_numpy(u), ref["u"]) < STEP_TOL
    assert ins.max_abs_divergence(u, sp) * (L / n[0]) < 1e-11


@pytest.mark.parametrize("kind", ["cavity", "xyper", "channel", "zwall", "zopen", "dctwalls", "dctsym", "dctopen"])
def test_rk44_with_fourier_directions_in_the_direct_solver(ins, oracle, kind):
    """Walls / open sides in one or two directions, the others periodic and uniform: the native stage loop with the direct solver whose
    periodic directions run in Fourier modes (divergence formed inside its x pass where x is periodic) against the oracle."""
    o = oracle
    so = _zperiodic(o, kind)
    sp = mirror(ins, so, o)
    ps_h, ps_d = o.psolver_direct(so), ins.psolver_direct(sp)
    g = so.grid
    u0 = o.project(o.apply_bc_u(0.1 * fx.randn_field(g.N + (3,), 21), 0.0, so), so, ps_h)
    o.apply_bc_u_(u0, 0.0, so)
    ref = o.solve_unsteady(so, (0.0, 4e-3), u0, psolver=ps_h, dt=2e-3)
    (u, _, _), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 4e-3), ustart=ins.from_numpy(sp, u0), psolver=ps_d, Δt=2e-3)
    assert rell2(ins.to_numpy(u), ref["u"]) < STEP_TOL


# ------------------------------------------------------------------ psolver_direct on strongly stretched grids
def _smooth_wall_field(o, so):
    g = so.grid
    D = g.D
    X = [g.xp[a].reshape([-1 if b == a else 1 for b in range(D)]) for a in range(D)]
    u = np.zeros(g.N + (D,), order="F")
    u[..., 0] = np.sin(np.pi * X[0]) * np.cos(np.pi * X[1]) + 0.3 * np.cos(2 * np.pi * X[0])
    u[..., 1] = np.cos(np.pi * X[0]) * np.sin(2 * np.pi * X[1])
    return o.apply_bc_u(u, 0.0, so)


def test_direct_solver_keeps_physical_modes_on_strongly_stretched_grid(ins, oracle):
    """tanh grid with h_min / L ~ 3e-7 (λmax = 4/h_min² ~ 7e12): a magnitude threshold of 1e-10·λmax·D on |λx+λy| sits far ABOVE the lowest
    physical eigenvalues (π²/L² ~ 9.9) and used to drop 128 modes (relative error 0.99 against the oracle's sparse LU of the bordered system,
    pressure.jl:133-140).  Only the one null mode may be dropped.  Tolerance 1e-6: cond(L) ~ 7e11, both solvers carry ~eps·cond
    (observed 5e-8 between a numpy fast diagonalisation and the LU)."""
    o = oracle
    x = (o.tanh_grid(0.0, 1.0, 64, 7.0), o.tanh_grid(0.0, 1.0, 48, 6.5))
    bc = (o.DirichletBC(), o.DirichletBC())
    so = o.make_setup(x, (bc, bc), Re=1000.0)
    sp = mirror(ins, so, o)
    g = so.grid
    ip = tuple(slice(lo, hi) for lo, hi in g.Ip)
    u_h = _smooth_wall_field(o, so)
    f = o.scalewithvolume(o.divergence(u_h, so), so)
    want = o.poisson(o.psolver_direct(so), f)
    solver = ins.psolver_direct(sp)
    got = ins.to_numpy(ins.poisson(solver, ins.from_numpy(sp, f)))
    w, q = want[ip] - want[ip].mean(), got[ip] - got[ip].mean()
    assert rell2(q, w) < 1e-6
    # and the projection really removes the divergence of the large scales
    ratio = _flux_imbalance_ratio(ins, sp, solver, u_h, ip)
    print(f"strongly stretched tanh grid: flux imbalance after / before project! = {ratio:.3e}")
    assert ratio < 1.2e-5  # 10 x the observed 1.2e-6 (cond(L) ~ 7e11); a partial regression of the null-mode rule (dropped low modes: O(1)) cannot pass


def _flux_imbalance_ratio(ins, sp, solver, u_h, ip):
    """Σ|Ω div u| after project! over the same sum before (the volume-scaled divergence is the quantity the solver sees; the plain
    max|div u| is dominated by the 1/h of the thinnest wall cells).  The oracle's LU gives 7e-13 / 3e-12 at N = 128 / 256 on the cosine grid;
    dropped low modes give O(1)."""
    u = ins.from_numpy(sp, u_h)
    before = np.abs(ins.to_numpy(ins.scalewithvolume(ins.divergence(u, sp), sp))[ip]).sum()
    ins.project_(u, sp, solver, ins.scalarfield(sp))
    after = np.abs(ins.to_numpy(ins.scalewithvolume(ins.divergence(u, sp), sp))[ip]).sum()
    return after / before


def test_direct_solver_cosine_grid_1024_projection(ins, oracle):
    """The reference's cosine grid at N = 1024 in 2-D (λmax = 4/h_min² ~ 1e11..1e12: the old threshold 1e-10·λmax·D = 25..150 dropped the modes
    (1,0), (0,1), (1,1) at 9.87, 9.87, 19.7).  The sparse LU oracle is too slow at this size (72 s at 512²), so the check is what the oracle's OWN
    operators say about the device result: L p = f - mean(f) with the numpy Laplacian (L1 norms), and the flux imbalance after project!."""
    o = oracle
    N = 1024
    x = (o.cosine_grid(0.0, 1.0, N), o.cosine_grid(0.0, 1.0, N))
    bc = (o.DirichletBC(), o.DirichletBC())
    so = o.make_setup(x, (bc, bc), Re=1000.0)
    sp = mirror(ins, so, o)
    g = so.grid
    ip = tuple(slice(lo, hi) for lo, hi in g.Ip)
    u_h = _smooth_wall_field(o, so)
    f = o.scalewithvolume(o.divergence(u_h, so), so)
    solver = ins.psolver_direct(sp)
    p = ins.to_numpy(ins.poisson(solver, ins.from_numpy(sp, f)))
    res = o.laplacian(o.apply_bc_p(p, 0.0, so), so)[ip] - (f[ip] - f[ip].mean())
    r1 = np.abs(res).sum() / np.abs(f[ip]).sum()
    r2 = _flux_imbalance_ratio(ins, sp, solver, u_h, ip)
    print(f"cosine grid 1024^2: |L p - f|_1 / |f|_1 = {r1:.3e}, flux imbalance after / before project! = {r2:.3e}")
    assert r1 < 1.3e-6  # 10 x the observed 1.3e-7 (cond(L) ~ 1e10..1e11); dropped low modes: O(1)
    assert r2 < 1.3e-6
    assert np.abs(p[ip]).max() > 1e-2  # the pressure of this smooth field lives in the first cosine modes (0.29 at N = 256)


# ------------------------------------------------------------------ masked grids: in-register pressure correction (config 5 path)
def _numpy_fast_diagonalisation(o, so):
    """The oracle's direct solve of `laplacian_mat` (pressure.jl:101-154, bordered when singular) for tensor-product grids by per-direction
    eigen-decompositions in numpy — the sparse LU of oracle.psolver_direct is too slow beyond ~20k unknowns.  Independent of the device code
    (dense numpy), and pinned to that LU on a small box by the caller."""
    g = so.grid
    D = g.D
    ip = tuple(slice(lo, hi) for lo, hi in g.Ip)
    L = o.laplacian_mat(so).toarray() if int(np.prod(g.Np)) <= 1500 else None
    V, lam = [], []
    for a in range(D):
        lo, hi = g.Ip[a]
        n = hi - lo
        # 1-D factor Tα by probing the oracle's laplacian! with unit vectors along direction a (everything else one volume wide is not
        # available, so build it from the definition: Ω/Δα ((p₊-p)/Δu[I] - (p-p₋)/Δu[I-1]) with apply_bc_p! ghosts, boundary branches of operators.jl:328-350)
        T = np.zeros((n, n))
        bl, br = so.boundary_conditions[a]
        du = g.dxu[a]
        for i in range(n):
            I = lo + i
            cr, cl = 1.0 / du[I], 1.0 / du[I - 1]
            first, last = i == 0, i == n - 1
            right = left = True
            gl = gr = None
            if first and isinstance(bl, o.PressureBC):
                pass
            elif last and isinstance(br, o.PressureBC):
                pass
            elif first and isinstance(bl, o.DirichletBC):
                left = False
            elif last and isinstance(br, o.DirichletBC):
                right = False
            else:
                if first:
                    gl = n - 1 if isinstance(bl, o.PeriodicBC) else i
                if last:
                    gr = 0 if isinstance(br, o.PeriodicBC) else i
            if right:
                T[i, i] -= cr
                j = i + 1 if not last else gr
                if j is not None:
                    T[i, j] += cr
            if left:
                T[i, i] -= cl
                j = i - 1 if not first else gl
                if j is not None:
                    T[i, j] += cl
        dm = 1.0 / np.sqrt(g.dx[a][lo:hi])
        l, W = np.linalg.eigh(dm[:, None] * T * dm[None, :])
        V.append(dm[:, None] * W)
        lam.append(l)
    singular = not any(isinstance(b, o.PressureBC) for side in so.boundary_conditions for b in side)
    lam = [l.copy() for l in lam]
    if singular:
        for l in lam:
            l[np.argmin(np.abs(l))] = 0.0
    S = lam[0].reshape(-1, 1, 1) + lam[1].reshape(1, -1, 1) + (lam[2].reshape(1, 1, -1) if D == 3 else 0.0)
    if D == 2:
        S = S[..., 0]
    sub = "abc"[:D]

    def psolve_(p):
        f = p[ip].copy()
        if singular:
            f -= f.mean()
        q = f
        for a in range(D):
            q = np.moveaxis(np.tensordot(V[a].T, q, axes=([1], [a])), 0, a)
        with np.errstate(divide="ignore", invalid="ignore"):
            q = np.where(S == 0.0, 0.0, q / S)
        for a in range(D):
            q = np.moveaxis(np.tensordot(V[a], q, axes=([1], [a])), 0, a)
        if singular:
            q -= q.mean()
        p[ip] = q
        return p

    return psolve_


@pytest.mark.parametrize("geom", ["cavity", "channel", "allwalls"])
@pytest.mark.parametrize("method", ["RK44", "Wray3"])
@pytest.mark.parametrize("nx", [64, 72, 136])  # 64: the 62-wide stage kernel; 72, 136: the 64-wide one (csrc/ins_flux64m.hip; a full + a partial wavefront, three wavefronts)
def test_masked_inkernel_correction_matches_oracle(ins, oracle, geom, method, nx):
    """Mid-size (nx x 48 x 32) stretched boxes with Dirichlet / Periodic sides and the direct solver: stages >= 2 read the previous stage's
    uncorrected u* and its pressure and apply the projection's gradient-subtract in registers on the degrees of freedom (stage kernels
    with CORR = 3; csrc/ins_rk.hip).  Three steps against the oracle's stage loop (step_explicit_runge_kutta.jl:17-50 with
    pressure.jl:69-82 after every stage; the oracle's direct solver in its fast-diagonalisation form, pinned above), and against the same
    library with the correction left to project! (INS_DISABLE_INKERNEL_CORR)."""
    from ins_amd import _lib

    o = oracle
    lid = (1.0, 0.2, 0.0)
    Do, Po, Dp, Pp = o.DirichletBC, o.PeriodicBC, ins.DirichletBC, ins.PeriodicBC
    if geom == "cavity":  # examples/LidDrivenCavity3D.jl: cosine x, y with a moving lid, periodic z
        x = (o.cosine_grid(0.0, 1.0, nx), o.cosine_grid(0.0, 1.0, 48), np.linspace(-0.2, 0.2, 33))
        bo = ((Do(), Do()), (Do(), Do(lid)), (Po(), Po()))
        bp = ((Dp(), Dp()), (Dp(), Dp(lid)), (Pp(), Pp()))
    elif geom == "channel":  # periodic x and z, tanh walls in y (Fourier x/z inside the direct solver)
        x = (np.linspace(0.0, 2.0, nx + 1), o.tanh_grid(0.0, 1.0, 48, 1.5), np.linspace(0.0, 1.0, 33))
        bo = ((Po(), Po()), (Do(), Do()), (Po(), Po()))
        bp = ((Pp(), Pp()), (Dp(), Dp()), (Pp(), Pp()))
    else:  # walls everywhere, stretched in all three directions
        x = (o.tanh_grid(0.0, 1.0, nx, 1.2), o.cosine_grid(0.0, 1.0, 48), o.tanh_grid(0.0, 0.5, 32, 1.1))
        bo = ((Do(), Do()), (Do(), Do(lid)), (Do(), Do()))
        bp = ((Dp(), Dp()), (Dp(), Dp(lid)), (Dp(), Dp()))
    so = o.make_setup(x, bo, Re=200.0)
    sp = ins.Setup(x=x, boundary_conditions=bp, Re=200.0)
    g = so.grid
    pso, psp = _numpy_fast_diagonalisation(o, so), ins.psolver_direct(sp)
    X = [g.xp[a].reshape([-1 if b == a else 1 for b in range(3)]) for a in range(3)]
    Lz = x[2][-1] - x[2][0]
    u0 = np.zeros(g.N + (3,), order="F")
    u0[..., 0] = 0.3 * np.sin(np.pi * X[0]) * np.cos(2 * np.pi * X[1]) * np.cos(2 * np.pi * X[2] / Lz)
    u0[..., 1] = -0.2 * np.cos(np.pi * X[0]) * np.sin(np.pi * X[1]) + 0 * X[2]
    u0[..., 2] = 0.1 * np.sin(2 * np.pi * X[0]) * np.sin(np.pi * X[1]) * np.sin(2 * np.pi * X[2] / Lz)
    u0 = o.apply_bc_u(u0, 0.0, so)
    u0 = o.project(u0, so, pso)
    o.apply_bc_u_(u0, 0.0, so)
    m, mo = getattr(ins.RKMethods, method)(), getattr(o, method)()
    dt = 0.5 * o.get_cfl_timestep(u0, so)
    want = o.solve_unsteady(so, (0.0, 3 * dt), u0, method=mo, psolver=pso, dt=dt)["u"]
    mask = np.zeros(g.N + (3,), dtype=bool)
    for a in range(3):
        mask[tuple(slice(max(lo_ - 1, 0), min(hi_ + 1, n_)) for (lo_, hi_), n_ in zip(g.Iu[a], g.N)) + (a,)] = True
    outs = {}
    for key, opts in (("corr", {}), ("project", {"INS_DISABLE_INKERNEL_CORR": 1})):
        with _lib.options(**opts):
            (u, _, t), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 3 * dt), ustart=ins.from_numpy(sp, u0), method=m, psolver=psp, Δt=dt)
            outs[key] = ins.to_numpy(u)
        assert rell2(outs[key][mask], want[mask]) < STEP_TOL, key
    assert rell2(outs["corr"][mask], outs["project"][mask]) < 1e-12


@pytest.mark.parametrize("n", [(64, 32), (128, 128), (256, 64), (16, 512), (64, 64), (32, 16)])  # the last two and the first: one-launch solves
@pytest.mark.parametrize("method", ["RK44", "Wray3", "SSP33"])
def test_fused_2d_stage_loop_with_in_register_correction(ins, oracle, n, method):
    """2-D periodic power-of-two boxes (the fused path: flux-form stage kernel + own passes): stages >= 2 read the previous stage's uncorrected u* and its
    unpadded pressure and apply u = u* - ∇p in registers (k_flux2d<…, CORR>, rows and columns through periodic images; 256 columns = five wavefront windows),
    the projection between two stages only solves.  Against the oracle's step (step_explicit_runge_kutta.jl:4-59) and against the loop with the
    gradient-subtract pass between the stages (INS_DISABLE_CORR2D)."""
    from ins_amd import _lib

    o = oracle
    so = fx.setup_periodic(o, n, D=2, Re=800.0)
    sp = mirror(ins, so, o)
    pso, psp = o.psolver_spectral(so), ins.psolver_spectral(sp)
    u0 = o.random_field(so, kp=3, seed=9)
    mo = getattr(o, method)()
    st = dict(setup=so, psolver=pso, u=u0.copy(order="F"), t=0.0, n=0)
    oc = o.ode_method_cache(mo, so)
    for _ in range(3):
        st = o.timestep_(mo, st, 2e-3, oc)
    m = getattr(ins.RKMethods, method)()

    def run(**opts):
        with _lib.options(**opts):
            cache = ins.ode_method_cache(m, sp, psp)
            s = ins.create_stepper(m, setup=sp, psolver=psp, u=ins.from_numpy(sp, u0), t=0.0)
            s = ins.timesteps_(m, s, 2e-3, 2, cache=cache)
            s = ins.timestep_(m, s, 2e-3, cache=cache)
            return ins.to_numpy(s.u), float(ins.max_abs_divergence(s.u, sp))

    got, div = run()
    ref, _ = run(INS_DISABLE_CORR2D=1)
    assert rell2(got, st["u"]) < STEP_TOL
    assert relmax(got, ref) < 1e-12
    assert div * (1.0 / max(n)) < 1e-12
