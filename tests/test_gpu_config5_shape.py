"""Parity at the launch geometry config 5 (LidDrivenCavity3D 256^3, stretched grid + Dirichlet sides, direct solver) really runs — VERDICT r02, weak item 2.

At 256^3 the 64-wide masked stage kernel (csrc/ins_flux64m.hip) runs four wavefronts side by side, several y tiles per XCD slot and 32-plane z-chunks; the
folded divergence (k_div_to_pI_fold), the half-size folded GEMMs, the cosine-transform z pass and the 2 x 2-image unfold (k_unfold_grad3) see their full-size
shapes.  Until now these kernels were oracle-checked at nx in {64, 72, 136} x 48 x 32 only.  Three kinds of checks, all through the C ABI:

  (a) full size, device A/B: one RK44 step of the default path against the reference's kernel sequence on the generic kernels (INS_DISABLE_FUSED_RK +
      INS_DISABLE_FAST3D + INS_DISABLE_FDM_FUSED; step_explicit_runge_kutta.jl:17-50, pressure.jl:69-82, operators.jl:647-690), same arrays, same solver
      handle: <= 1e-12 relative max-norm; and against the same sequence with a solver that uses NONE of the structured shortcuts (dense eigenvector
      GEMMs in all three directions: no fold, no Fourier / cosine pass) at the solver's own conditioning;
  (b) mid-size boxes that force the bench tile shapes (256 x 72 x 40, 256 x 24 x 136: four wavefronts side by side, nty_local > 1, 32-plane chunks with a
      ragged tail) against the numpy oracle's stage loop with its direct solver in fast-diagonalisation form (pinned to the sparse LU in
      tests/test_oracle_pinning.py), <= 1e-10;
  (c) the wall-bounded extended loops (temperature equation, Smagorinsky closure) at 256^3 / 256 x 128 x 256: fused loop against the reference's kernel
      sequence on the device (INS_DISABLE_EXT_FUSED), <= 1e-12 relative max-norm.
"""
import numpy as np
import pytest

from tests.test_gpu_parity import STEP_TOL, _numpy_fast_diagonalisation, ins, rell2  # noqa: F401  (ins: the module-scoped fixture)

pytestmark = pytest.mark.gpu

LID = (1.0, 0.2, 0.0)
GENERIC = dict(INS_DISABLE_FUSED_RK=1, INS_DISABLE_FAST3D=1, INS_DISABLE_FDM_FUSED=1)
DENSE_SOLVER = dict(INS_DISABLE_FDM_FOLD=1, INS_DISABLE_FDM_ZFFT=1, INS_DISABLE_FDM_ZDCT=1, INS_DISABLE_FDM_XFFT=1, INS_DISABLE_FDM_XYFFT=1)


def _tanh_grid(a, b, n, g=1.5):
    s = np.linspace(-1, 1, n + 1)
    return a + (b - a) * (1 + np.tanh(g * s) / np.tanh(g)) / 2


def _geometry(ins, name, n):
    """The grids of tools/cavity_prof.py, walls_prof.py and channel_prof.py (what DESIGN §5's config-5 table was measured on)."""
    D, P = ins.DirichletBC, ins.PeriodicBC
    if name == "cavity":  # examples/LidDrivenCavity3D.jl:26-40: cosine x cosine x periodic, lid (1, 0.2, 0) on y-right
        return (ins.cosine_grid(0.0, 1.0, n[0]), ins.cosine_grid(0.0, 1.0, n[1]), np.linspace(-0.2, 0.2, n[2] + 1)), ((D(), D()), (D(), D(LID)), (P(), P()))
    if name == "allwalls":  # uniform box, Dirichlet on six sides (the cosine-transform z pass)
        return tuple(np.linspace(0.0, 1.0, ni + 1) for ni in n), ((D(), D()), (D(), D(LID)), (D(), D()))
    if name == "channel":  # periodic x and z, tanh walls in y
        return (np.linspace(0.0, 4.0, n[0] + 1), _tanh_grid(0.0, 1.0, n[1]), np.linspace(0.0, 2.0, n[2] + 1)), ((P(), P()), (D(), D()), (P(), P()))
    raise KeyError(name)


def _wall_field(ins, sp, ps):
    """A smooth, non-trivial, projected start field with the boundary data applied (the lid-driven cavity itself starts from rest)."""
    Lz = float(sp.grid.x[2][-2] - sp.grid.x[2][1])

    def f(a, x, y, z):
        if a == 0:
            return 0.3 * np.sin(np.pi * x) * np.cos(2 * np.pi * y) * np.cos(2 * np.pi * z / Lz)
        if a == 1:
            return -0.2 * np.cos(np.pi * x) * np.sin(np.pi * y) + 0 * z
        return 0.1 * np.sin(2 * np.pi * x) * np.sin(np.pi * y) * np.sin(2 * np.pi * z / Lz)

    return ins.velocityfield(sp, f, 0.0, psolver=ps)


def _one_step(ins, sp, ps, u0, dt, **kw):
    m = ins.RKMethods.RK44()
    cache = ins.ode_method_cache(m, sp, ps)
    st = ins.create_stepper(m, setup=sp, psolver=ps, u=ins.copyfield(u0), t=0.0, **{k: v for k, v in kw.items() if k == "temp"})
    st = ins.timestep_(m, st, dt, cache=cache, **{k: v for k, v in kw.items() if k == "θ"})
    return st


def _dof_mask(sp, like):
    """Degrees of freedom grown by one ghost layer (what later kernels read); volumes outside it are left as allocated by either path."""
    import torch

    g = sp.grid
    mask = torch.zeros(like.shape, dtype=torch.bool, device=like.device)
    for a in range(3):
        sl = tuple(slice(max(lo - 1, 0), min(hi + 1, n)) for (lo, hi), n in zip(g.Iu[a], g.N))
        mask[sl + (a,)] = True
    return mask


# ------------------------------------------------------------------------------------------------ (a) full size, device A/B
@pytest.mark.parametrize("name,n", [("cavity", (256, 256, 256)), ("allwalls", (256, 256, 256)), ("channel", (256, 128, 256))])
def test_config5_full_size_fused_vs_reference_order(ins, name, n):
    import torch

    from ins_amd import _lib

    x, bc = _geometry(ins, name, n)
    sp = ins.Setup(x=x, Re=1000.0, boundary_conditions=bc)
    ps = ins.psolver_direct(sp)
    u0 = _wall_field(ins, sp, ps)
    dt = 0.5 * float(ins.get_cfl_timestep_(None, u0, sp))
    mask = _dof_mask(sp, u0)
    a = _one_step(ins, sp, ps, u0, dt).u
    assert bool(torch.isfinite(a).all())
    with _lib.options(**GENERIC):
        b = _one_step(ins, sp, ps, u0, dt).u
    scale = float(b[mask].abs().max())
    assert scale > 0.05
    assert float((a - b)[mask].abs().max()) < 1e-12 * scale
    assert float((b - u0)[mask].abs().max()) > 1e-6 * scale  # the step did something
    if name == "channel":  # (the lid (1, 0.2, 0) has a normal component: the bordered system then has no divergence-free solution, pressure.jl:133-140)
        assert float(ins.max_abs_divergence(a, sp)) * float(np.min(sp.grid.Δ[1][1:-1])) < 1e-9
    del b
    # the same sequence with a solver that takes no structured shortcut (dense eigenvector GEMMs in every direction): a different
    # factorisation of the same matrix, so the two agree at the solver's conditioning (observed: see DESIGN §4), not at rounding level
    with _lib.options(**DENSE_SOLVER):
        ps2 = ins.psolver_direct(sp)
    with _lib.options(**GENERIC):
        c = _one_step(ins, sp, ps2, u0, dt).u
    err = float((a - c)[mask].abs().max()) / scale
    print(f"config5 {name} {n}: fused vs dense-solver reference order: {err:.3e}")
    assert err < 1e-10  # observed 4.9e-12 (cavity: cond ~ 3e8), 1.2e-14 (all-walls), 1.2e-15 (channel)
    torch.cuda.synchronize()
    del a, c, ps, ps2, sp
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------ (b) bench tile shapes against the oracle
@pytest.mark.parametrize("n,zc", [((256, 72, 40), 32), ((256, 24, 136), 0), ((256, 40, 72), 32)])
@pytest.mark.parametrize("name", ["cavity", "allwalls", "channel"])
@pytest.mark.parametrize("method", ["RK44", "Wray3"])
def test_config5_bench_tile_shapes_match_oracle(ins, oracle, name, n, zc, method):
    """256-wide rows (four wavefronts side by side), more y tiles than XCD slots, 32-plane chunks (forced where the box has fewer than 128 planes; ragged
    tails 32 + 8, 4 x 32 + 8, 2 x 32 + 8): three steps through the stage loop with the in-register correction on masked grids against the oracle's loop."""
    from ins_amd import _lib

    o = oracle
    x, bc = _geometry(ins, name, n)
    bo = tuple(tuple(getattr(o, type(b).__name__)(*((b.u,) if getattr(b, "u", None) is not None else ())) for b in pair) for pair in bc)
    so = o.make_setup(x, bo, Re=200.0)
    sp = ins.Setup(x=x, boundary_conditions=bc, Re=200.0)
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
    with _lib.options(INS_FLUX64M_ZC=zc):
        (u, _, t), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 3 * dt), ustart=ins.from_numpy(sp, u0), method=m, psolver=psp, Δt=dt)
        got = ins.to_numpy(u)
    assert rell2(got[mask], want[mask]) < STEP_TOL
    with _lib.options(INS_FLUX64M_ZC=zc, **GENERIC):
        (u, _, t), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 3 * dt), ustart=ins.from_numpy(sp, u0), method=m, psolver=psp, Δt=dt)
        ref = ins.to_numpy(u)
    assert rell2(got[mask], ref[mask]) < 1e-12


# ------------------------------------------------------------------------------------------------ (c) wall-bounded extended loops at full size
@pytest.mark.parametrize("name,n,what", [("allwalls", (256, 256, 256), "temperature"), ("allwalls", (256, 256, 256), "closure"),
                                         ("channel", (256, 128, 256), "closure"), ("cavity", (256, 256, 256), "closure")])
def test_wall_bounded_extended_loops_full_size(ins, name, n, what):
    """The tiled extended stage loop (64-wide masked stage kernel taking gravity / the closure force and leaving u·diffusion(u), one temperature kernel per
    stage, the one-kernel closure force in its generalised forms) against the reference's kernel sequence on the device (INS_DISABLE_EXT_FUSED), 256^3."""
    import torch

    from ins_amd import _lib

    x, bc = _geometry(ins, name, n)
    D, S = ins.DirichletBC, ins.SymmetricBC
    temperature = None
    if what == "temperature":  # examples/RayleighTaylor3D.jl shape: Symmetric temperature sides, gravity along z
        temperature = ins.temperature_equation(Pr=0.71, Ra=1e6, Ge=1.0, boundary_conditions=((S(), S()),) * 3, gdir=2)
    sp = ins.Setup(x=x, Re=1000.0 if temperature is None else None, boundary_conditions=bc, temperature=temperature)
    if what == "closure":
        sp.closure_model = ins.smagorinsky_closure(sp)
    ps = ins.psolver_direct(sp)
    u0 = _wall_field(ins, sp, ps)
    temp0 = None if temperature is None else ins.temperaturefield(sp, lambda x, y, z: 0.5 + 0.3 * np.sin(np.pi * x) * np.cos(np.pi * y) * np.cos(2 * np.pi * z))
    dt = 0.4 * float(ins.get_cfl_timestep_(None, u0, sp))
    kw = {}
    if temp0 is not None:
        kw["temp"] = temp0
    if what == "closure":
        kw["θ"] = 0.17
    mask = _dof_mask(sp, u0)

    def run():
        st = _one_step(ins, sp, ps, u0, dt, **({"temp": ins.copyfield(temp0)} if temp0 is not None else {}), **{k: v for k, v in kw.items() if k == "θ"})
        return st.u, st.temp

    a, ta = run()
    with _lib.options(INS_DISABLE_EXT_FUSED=1):
        b, tb = run()
    scale = float(b[mask].abs().max())
    assert scale > 0.05 and bool(torch.isfinite(a).all())
    assert float((a - b)[mask].abs().max()) < 1e-12 * scale
    if ta is not None:
        ip = tuple(slice(lo, hi) for lo, hi in sp.grid.Ip)
        assert float((ta - tb)[ip].abs().max()) < 1e-12 * float(tb[ip].abs().max())
        assert float((tb - temp0)[ip].abs().max()) > 0
    torch.cuda.synchronize()
    del a, b, ps, sp
    torch.cuda.empty_cache()
