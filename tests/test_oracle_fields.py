"""CPU checks of the oracle's restatement of the step-adjacent operators against closed forms and
discrete identities.  The reference's own tests hold no values for these operators (parity beyond these checks is "unpinned")."""
import numpy as np
import pytest

from tests import fixtures as fx


def tgv2d(o, n, Re=1e3):
    x = (np.linspace(0, 2 * np.pi, n + 1),) * 2
    s = o.make_setup(x, Re=Re)
    u = o.velocityfield(s, lambda a, x, y: (-np.sin(x) * np.cos(y) if a == 0 else np.cos(x) * np.sin(y)), doproject=False)
    return s, u


def test_vorticity_and_q_of_taylor_green_converge_at_second_order(oracle):
    o = oracle
    ev, eq = [], []
    for n in (32, 64):
        s, u = tgv2d(o, n)
        g = s.grid
        R = (slice(1, n + 1),) * 2
        X, Y = g.x[0][1:][:, None], g.x[1][1:][None, :]  # ω[I] lives at the corner (x[i+1], y[j+1])
        ev.append(np.abs(o.vorticity(u, s)[R] - (-2 * np.sin(X) * np.sin(Y))[R]).max())
        Xp, Yp = g.xp[0][1:-1][:, None], g.xp[1][1:-1][None, :]
        Qe = -((np.cos(Xp) * np.cos(Yp)) ** 2 - (np.sin(Xp) * np.sin(Yp)) ** 2)
        eq.append(np.abs(o.Qfield(u, s)[1:-1, 1:-1] - Qe).max())
        G = o.gradu(u, s)
        assert np.abs(G[..., 0, 0] + G[..., 1, 1]).max() < 1e-13  # trace = discrete divergence of an exactly solenoidal field
    assert 3.7 < ev[0] / ev[1] < 4.3 and 3.7 < eq[0] / eq[1] < 4.3


def test_interpolations(oracle):
    o = oracle
    s, u = tgv2d(o, 48)
    g = s.grid
    up = o.interpolate_u_p(u, s)
    ex = -np.sin(g.xp[0][1:-1])[:, None] * np.cos(g.xp[1][1:-1])[None, :]
    assert np.abs(up[1:-1, 1:-1, 0] - ex).max() < 3e-3
    w = o.vorticity(u, s)
    wp = o.interpolate_w_p(w, s)
    ex = -2 * np.sin(g.xp[0][1:-1])[:, None] * np.sin(g.xp[1][1:-1])[None, :]
    assert np.abs(wp[1:-1, 1:-1] - ex).max() < 2e-2


def test_temperature_operators_identities(oracle):
    o = oracle
    n = 32
    s0, u = tgv2d(o, n)
    bcT = ((o.PeriodicBC(), o.PeriodicBC()),) * 2
    T = o.temperature_equation(Pr=0.71, Ra=1e6, Ge=0.1, boundary_conditions=bcT)
    s = o.make_setup_ext(s0.grid.x[0][1:-1][None].repeat(2, 0), temperature=T)
    assert s.Re == pytest.approx(1 / T.a1)
    h2 = (2 * np.pi / n) ** 2
    temp = o.temperaturefield(s, lambda x, y: 1 + 0.3 * np.sin(x) * np.cos(2 * y))
    c = o.convection_diffusion_temp(u, temp, s)
    assert abs(c[1:-1, 1:-1].sum() * h2) < 1e-13  # flux form: conservative on a periodic box
    F = o.gravity(np.full_like(temp, 2.0), s)
    assert np.allclose(F[1:-1, 1:-1, 1], 2.0 * T.a2) and np.all(F[..., 0] == 0)
    # Σ Ω u·D(u) = -Σ Ω 2ν S:S on a periodic box with div u = 0 (both sides discrete)
    d = o.dissipation(u, s)[1:-1, 1:-1].sum() * h2 / (s.Re * T.a1 / T.gamma)
    e = o.dissipation_from_strain(u, s)[1:-1, 1:-1].sum() * h2
    assert d == pytest.approx(-e, rel=1e-12)
    assert e == pytest.approx(2 / s.Re * (2 * np.pi) ** 2 * 0.5, rel=5e-3)  # ∫ 2ν S:S of the TGV


def test_smagorinsky_closure_is_dissipative(oracle):
    o = oracle
    s, u = tgv2d(o, 24)
    sf = o.smagorinsky_closure(s)(u, 0.17)
    assert (sf[1:-1, 1:-1, :] * u[1:-1, 1:-1, :]).sum() < 0
    s3 = fx.setup_periodic(o, (8, 8, 8))
    u3 = o.random_field(s3, kp=2, seed=1)
    sf3 = o.smagorinsky_closure(s3)(u3, 0.1)
    assert (sf3[1:-1, 1:-1, 1:-1, :] * u3[1:-1, 1:-1, 1:-1, :]).sum() < 0
    lam = o.eig2field(u3, s3)[1:-1, 1:-1, 1:-1]
    Q = o.Qfield(u3, s3)[1:-1, 1:-1, 1:-1]
    assert np.isfinite(lam).all() and np.isfinite(Q).all()


def test_steppers_with_temperature_keep_mean_temperature(oracle):
    """Periodic box, no dissipation heating: both steppers conserve Σ T exactly (flux form) and agree to O(Δt³)."""
    o = oracle
    n = 16
    x = (np.linspace(0, 2 * np.pi, n + 1),) * 2
    bcT = ((o.PeriodicBC(), o.PeriodicBC()),) * 2
    T = o.temperature_equation(Pr=0.71, Ra=1e4, Ge=0.1, boundary_conditions=bcT, dodissipation=False)
    s = o.make_setup_ext(x, temperature=T)
    ps = o.psolver_spectral(s)
    u0 = o.velocityfield(s, lambda a, x, y: (-np.sin(x) * np.cos(y) if a == 0 else np.cos(x) * np.sin(y)), psolver=ps)
    t0 = o.temperaturefield(s, lambda x, y: 1 + 0.3 * np.sin(x) * np.cos(2 * y))
    cache = o.ode_method_cache_ext(o.RK44(), s)
    out = []
    for step in (lambda st: o.timestep_ext_(o.RK44(), st, 1e-2, cache), lambda st: o.timestep_lmwray3_ext_(st, 1e-2, cache)):
        st = dict(setup=s, psolver=ps, u=u0.copy(order="F"), temp=t0.copy(order="F"), t=0.0, n=0)
        for _ in range(3):
            st = step(st)
        assert st["temp"][1:-1, 1:-1].mean() == pytest.approx(t0[1:-1, 1:-1].mean(), abs=1e-14)
        out.append(st)
    assert np.abs(out[0]["u"] - out[1]["u"]).max() < 1e-5 and np.abs(out[0]["temp"] - out[1]["temp"]).max() < 1e-5
    assert np.abs(out[0]["u"] - u0).max() > 1e-3  # buoyancy did act


def test_spectrum_of_a_single_mode_lands_in_its_shell(oracle):
    """One Fourier mode of wavenumber (3, 0, 0) with amplitude A: E = A²/4 per component... all of it in the shell κ = 3 (3-D linear bins)."""
    o = oracle
    n = 16
    s = fx.setup_periodic(o, (n, n, n))
    u = o.vectorfield(s)
    X = s.grid.xp[0][:, None, None]
    u[..., 1] = 0.7 * np.cos(2 * np.pi * 3 * X) + 0 * u[..., 1]
    ehat, kap = o.observespectrum(u, s)
    # fft amplitude of A cos at +k: A N³/2 -> |û|²/(2 N⁶) = A²/8 in the retained (non-negative) half
    i3 = list(kap).index(3)
    assert ehat[i3] == pytest.approx(0.7**2 / 8, rel=1e-12)
    assert np.abs(np.delete(ehat, i3)).max() < 1e-28
    inds, kap2, K = o.spectral_stuff(s)
    assert K == (8, 8, 8) and kap2[-1] == 7 and all(len(i) > 0 for i in inds)
