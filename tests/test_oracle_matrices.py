"""The reference's matrix tests, re-run with the oracle in the role of the package (CPU, no GPU).

  test/matrices.jl:19-51      BC matrices / divergence / pressure gradient / diffusion: stencil operator == index-assembled sparse matrix
                              on the 11x7x5 mixed-BC fixture (Periodic x Dirichlet|Pressure x Symmetric) — here also on the 16-cell stretched fixtures
  test/operators.jl:90-105    laplacian! == laplacian_mat (= P' Ω M Bu G Bp P of the assembled factors), negative semi-definite

The matrices come from oracle/ins_matrices.py, a restatement of src/matrices.jl:1-555 that builds every triplet from index ranges and never calls the
stencil code of oracle/ins_oracle.py; so these are comparisons of two implementations, as in the reference.  The probed matrix of the stencil operators
(ins_oracle.laplacian_mat_probed) is kept as a third cross-check, and psolver_direct factorises the assembled matrix."""
import numpy as np
import pytest

from oracle import ins_matrices as m
from tests import fixtures as fx


def _flat(a):
    return np.asarray(a).reshape(-1, order="F")


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _periodic_stretchless(o):
    return fx.setup_periodic(o, (8, 6, 5))


FIXTURES = {"mixed": fx.setup_mixed, "setup2d": fx.setup2d, "setup3d": fx.setup3d, "periodic3d": _periodic_stretchless}
TOL = 1e-13  # `≈` in the reference is rtol = sqrt(eps); the two implementations here agree to rounding


@pytest.fixture(params=list(FIXTURES))
def case(request, oracle):
    so = FIXTURES[request.param](oracle)
    g = so.grid
    return oracle, so, fx.randn_field(g.N + (g.D,), 11), fx.randn_field(g.N, 12)


def test_bc_matrices(case):  # test/matrices.jl:19-29
    o, so, u, p = case
    Bu, Bp = m.bc_u_mat(so), m.bc_p_mat(so)
    hs = o._homogeneous(so)  # (the matrix holds the part that depends on the field; these fixtures have homogeneous data anyway)
    assert np.array_equal(Bu @ _flat(u), _flat(o.apply_bc_u(u, 0.0, hs)))
    assert np.array_equal(Bp @ _flat(p), _flat(o.apply_bc_p(p, 0.0, hs)))


def test_divergence_matrix(case):  # test/matrices.jl:31-37
    o, so, u, p = case
    div1 = m.divergence_mat(so) @ (m.bc_u_mat(so) @ _flat(u))
    div2 = o.divergence(o.apply_bc_u(u, 0.0, so), so)
    assert np.abs(div2).max() > 0 and _rel(div1, _flat(div2)) < TOL


def test_pressuregradient_matrix(case):  # test/matrices.jl:39-45
    o, so, u, p = case
    g1 = m.pressuregradient_mat(so) @ (m.bc_p_mat(so) @ _flat(p))
    g2 = o.pressuregradient(o.apply_bc_p(p, 0.0, so), so)
    assert np.abs(g2).max() > 0 and _rel(g1, _flat(g2)) < TOL


def test_diffusion_matrix(case):  # test/matrices.jl:47-53
    o, so, u, p = case
    d1 = m.diffusion_mat(so) @ (m.bc_u_mat(so) @ _flat(u))
    d2 = o.diffusion(o.apply_bc_u(u, 0.0, so), so, use_viscosity=False)
    assert np.abs(d2).max() > 0 and _rel(d1, _flat(d2)) < TOL


def test_laplacian_kernel_equals_assembled_matrix(case):  # test/operators.jl:90-105
    o, so, u, p = case
    g = so.grid
    ip = o._sl(g.Ip)
    p = o.apply_bc_p(p, 0.0, so)
    Lp = o.laplacian(p, so)
    assert float(np.sum((p * o.scalewithvolume(Lp, so))[ip])) <= 0  # negativity
    L = m.laplacian_mat(so)
    assert float(np.sum((_flat(Lp[ip]) - L @ _flat(p[ip])) ** 2)) == pytest.approx(0.0, abs=1e-12)
    # symmetric (pressure.jl:137 guard) and equal to the probed matrix of the stencil operators
    assert abs(L - L.T).max() <= 1e-12 * abs(L).max()
    assert abs(L - o.laplacian_mat_probed(so)).max() <= 1e-13 * abs(L).max()


def test_gradient_is_minus_divergence_transpose_in_matrix_form(case):
    """test/operators.jl:58-88 (D = -G') stated on the matrices: restricted to the degrees of freedom and weighted by the volumes of the velocity
    control volumes, Pu' Ωu G P = -(Pp' Ω M Pu)' (Dirichlet / Symmetric / Periodic sides; a PressureBC side adds boundary terms: skipped)."""
    o, so, u, p = case
    if any(isinstance(b, o.PressureBC) for pair in so.boundary_conditions for b in pair):
        pytest.skip("boundary terms at a PressureBC side")
    g = so.grid
    D = g.D
    Pp, Pu = m.pad_scalarfield_mat(so), m.pad_vectorfield_mat(so)
    import scipy.sparse as sp

    vols = []
    for a in range(D):
        v = np.ones(g.N)
        for b in range(D):
            w = g.dxu[b] if a == b else g.dx[b]
            v = v * np.asarray(w).reshape([-1 if d == b else 1 for d in range(D)])
        vols.append(_flat(v))
    Ou = sp.diags(np.concatenate(vols))
    Bu, Bp = m.bc_u_mat(so), m.bc_p_mat(so)
    GP = Pu.T @ Ou @ m.pressuregradient_mat(so) @ Bp @ Pp
    MP = Pp.T @ m.volume_mat(so) @ m.divergence_mat(so) @ Bu @ Pu
    assert abs(GP + MP.T).max() <= 1e-12 * abs(GP).max()


def test_direct_solver_factorises_the_assembled_matrix(oracle):
    """pressure.jl:117-154 on the assembled matrix: consistent right-hand side, bordered (singular) and definite (PressureBC) systems."""
    o = oracle
    for mk in (fx.setup2d, fx.setup_mixed):
        so = mk(o)
        g = so.grid
        ip = o._sl(g.Ip)
        L = m.laplacian_mat(so)
        f = o.scalarfield(so)
        rhs = fx.randn_field(g.Np, 5)
        definite = any(isinstance(b, o.PressureBC) for pair in so.boundary_conditions for b in pair)
        if not definite:
            rhs -= rhs.mean()
        f[ip] = rhs
        psol = o.psolver_direct(so)(f.copy())
        assert _rel(L @ _flat(psol[ip]), _flat(rhs)) < 1e-10
        if not definite:
            assert abs(psol[ip].sum()) < 1e-10 * np.abs(psol[ip]).sum()
