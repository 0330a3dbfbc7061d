"""Host logic: the explicit Butcher tableaux of RKMethods.jl satisfy the order conditions their names promise (s stages, order p),
are stored shifted as methods.jl:231-236 prescribes, and the implicit ones are rejected.  (GPU: a few of them step like the oracle.)"""
import numpy as np
import pytest

TABLEAUX = [("FE11", 1, 1), ("SSP22", 2, 2), ("SSP42", 4, 2), ("SSP33", 3, 3), ("SSP43", 4, 3), ("SSP104", 10, 4), ("rSSPs2", 2, 2), ("rSSPs3", 16, 3),
            ("Wray3", 3, 3), ("RK56", 6, 5), ("DOPRI6", 6, 5), ("Mid22", 2, 2), ("MTE22", 2, 2), ("Heun33", 3, 3), ("RK33C2", 3, 3), ("RK33P2", 3, 3),
            ("RK44", 4, 4), ("RK44C2", 4, 4), ("RK44C23", 4, 4), ("RK44P2", 4, 4), ("NSSP21", 2, 1), ("NSSP32", 3, 2), ("NSSP33", 3, 3), ("NSSP53", 5, 3)]


def unshift(m):
    s = len(m.b)
    return np.vstack([np.zeros((1, s)), m.A[:-1]]), m.b, np.concatenate([[0.0], m.c[:-1]])


@pytest.mark.parametrize("name,stages,order", TABLEAUX)
def test_order_conditions(name, stages, order):
    from ins_amd.time_steppers import RKMethods

    m = getattr(RKMethods, name)()
    A, b, c = unshift(m)
    assert len(b) == stages and np.allclose(np.triu(A), 0) and np.array_equal(m.A[-1], m.b) and m.c[-1] == 1.0
    cond = [b.sum() - 1]
    if order >= 2:
        cond.append(b @ c - 1 / 2)
    if order >= 3:
        cond += [b @ c**2 - 1 / 3, b @ (A @ c) - 1 / 6]
    if order >= 4:
        cond += [b @ c**3 - 1 / 4, (b * c) @ (A @ c) - 1 / 8, b @ (A @ c**2) - 1 / 12, b @ (A @ (A @ c)) - 1 / 24]
    if order >= 5:
        cond += [b @ c**4 - 1 / 5, b @ (A @ c**3) - 1 / 20, b @ (A @ (A @ (A @ c))) - 1 / 120]
    assert max(abs(v) for v in cond) < 1e-13


def test_family_parameters_and_rejections():
    from ins_amd.time_steppers import RKMethods, runge_kutta_method

    assert len(RKMethods.rSSPs2(5).b) == 5 and RKMethods.rSSPs2(5).r == 4
    with pytest.raises(ValueError):
        RKMethods.rSSPs2(1)
    with pytest.raises(ValueError):
        RKMethods.rSSPs3(5)
    with pytest.raises(NotImplementedError):  # CN22 (RKMethods.jl:479-485): implicit, legacy in the reference
        runge_kutta_method([[0, 0], [0.5, 0.5]], [0.5, 0.5], [0, 1])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["RK44P2", "DOPRI6", "SSP104", "NSSP33", "rSSPs3"])
def test_methods_step_like_the_oracle(oracle, name):
    """3 steps on a ragged periodic box and on the stretched Dirichlet box: native stage loop (k-basis when a diagonal entry of the shifted
    tableau vanishes, up to 16 stages) against the oracle driven with the same tableau."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ins_amd as ins
    from tests import fixtures as fx
    from tests.test_gpu_parity import GEOMS, mirror, rell2

    o = oracle
    m = getattr(ins.RKMethods, name)()
    mo = o.ExplicitRungeKuttaMethod(m.A.copy(), m.b.copy(), m.c.copy(), m.r)
    for geom in ("periodic3d", "dirichlet3d", "periodic2d"):
        so = GEOMS[geom](o)
        sp = mirror(ins, so, o)
        ps_h, ps_d = o.default_psolver(so), ins.default_psolver(sp)
        u0 = o.project(o.apply_bc_u(0.1 * fx.randn_field(so.grid.N + (so.grid.D,), 5), 0.0, so), so, ps_h)
        o.apply_bc_u_(u0, 0.0, so)
        ref = o.solve_unsteady(so, (0.0, 6e-3), u0, method=mo, psolver=ps_h, dt=2e-3)
        (u, _, t), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 6e-3), ustart=ins.from_numpy(sp, u0), method=m, psolver=ps_d, Δt=2e-3)
        assert rell2(ins.to_numpy(u), ref["u"]) < 1e-10
