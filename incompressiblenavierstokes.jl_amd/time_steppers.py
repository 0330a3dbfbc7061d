"""Explicit Runge-Kutta time stepping (time_steppers/methods.jl, RKMethods.jl,
step_explicit_runge_kutta.jl, time_stepper_caches.jl), host side."""
import ctypes as C
import os
from dataclasses import dataclass
from fractions import Fraction
from types import SimpleNamespace

import numpy as np

from . import _lib
from .operators import apply_bc_temp_, apply_bc_u_, convection_diffusion_temp_, dissipation_, momentum_
from .pressure import project_
from .setup import copyfield, scalarfield, vectorfield


@dataclass
class ExplicitRungeKuttaMethod:
    """methods.jl:184-190 (A and c already shifted, methods.jl:231-236)."""

    A: np.ndarray
    b: np.ndarray
    c: np.ndarray
    r: float = 0.0
    p_add_solve: bool = True


def runge_kutta_method(A, b, c, r=0.0, **kw):
    """methods.jl:219-240 (explicit tableaux only: the implicit branch is dead code in the reference)."""
    A = np.array(A, dtype=np.float64)
    b = np.array(b, dtype=np.float64)
    c = np.array(c, dtype=np.float64)
    s = A.shape[0]
    if not (A.shape == (s, s) and b.shape == (s,) and c.shape == (s,)):
        raise ValueError("A, b, and c must have the same sizes")
    if not np.allclose(np.triu(A), 0):
        raise NotImplementedError("implicit Runge-Kutta methods are legacy code in the reference (SURVEY.md §2)")
    A = np.vstack([A[1:, :], b[None, :]])
    c = np.concatenate([c[1:], [1.0]])
    return ExplicitRungeKuttaMethod(A, b, c, float(r), **kw)


class RKMethods:
    """Explicit tableaux of RKMethods.jl (same names)."""

    @staticmethod
    def FE11(**kw):  # RKMethods.jl:45-50
        return runge_kutta_method([[0.0]], [1.0], [0.0], **kw)

    @staticmethod
    def SSP22(**kw):  # :54-59
        A = np.array([[0, 0], [1, 0]], dtype=float)
        return runge_kutta_method(A, [0.5, 0.5], A.sum(1), **kw)

    @staticmethod
    def SSP42(**kw):  # :63-69
        t = 1 / 3
        A = np.array([[0, 0, 0, 0], [t, 0, 0, 0], [t, t, 0, 0], [t, t, t, 0]])
        return runge_kutta_method(A, [0.25] * 4, A.sum(1), **kw)

    @staticmethod
    def SSP33(**kw):  # :73-78
        A = np.array([[0, 0, 0], [1, 0, 0], [0.25, 0.25, 0]])
        return runge_kutta_method(A, [1 / 6, 1 / 6, 2 / 3], A.sum(1), **kw)

    @staticmethod
    def SSP43(**kw):  # :82-87
        s = 1 / 6
        A = np.array([[0, 0, 0, 0], [0.5, 0, 0, 0], [0.5, 0.5, 0, 0], [s, s, s, 0]])
        return runge_kutta_method(A, [s, s, s, 0.5], A.sum(1), **kw)

    @staticmethod
    def Wray3(**kw):  # :137-147
        a31 = 8 / 15 - 17 / 60
        A = np.array([[0, 0, 0], [8 / 15, 0, 0], [a31, 5 / 12, 0]])
        return runge_kutta_method(A, [a31, 0, 0.75], [0, 8 / 15, a31 + 5 / 12], **kw)

    @staticmethod
    def RK44(**kw):  # :515-521
        A = np.array([[0, 0, 0, 0], [0.5, 0, 0, 0], [0, 0.5, 0, 0], [0, 0, 1, 0]])
        return runge_kutta_method(A, [1 / 6, 1 / 3, 1 / 3, 1 / 6], A.sum(1), **kw)

    @staticmethod
    def RK44C2(**kw):  # :524-530
        A = np.array([[0, 0, 0, 0], [0.25, 0, 0, 0], [0, 0.5, 0, 0], [1, -2, 2, 0]])
        return runge_kutta_method(A, [1 / 6, 0, 2 / 3, 1 / 6], A.sum(1), **kw)

    @staticmethod
    def RK33P2(**kw):  # :506-512
        A = np.array([[0, 0, 0], [1 / 3, 0, 0], [-1, 2, 0]])
        return runge_kutta_method(A, [0, 0.75, 0.25], [0, 1 / 3, 1], **kw)


    # ---- the rest of the explicit tableaux of RKMethods.jl (published coefficients; rows of A below the diagonal) ----
    @staticmethod
    def _from_rows(rows, b, c=None, r=0.0, **kw):
        s = len(b)
        A = np.zeros((s, s))
        for i, row in enumerate(rows, start=1):
            A[i, : len(row)] = [float(Fraction(v)) for v in row]
        b = [float(Fraction(v)) for v in b]
        c = A.sum(1) if c is None else [float(Fraction(v)) for v in c]
        return runge_kutta_method(A, b, c, r, **kw)

    @staticmethod
    def _from_shu_osher(alpha, beta, r, **kw):
        """Butcher form of a Shu-Osher pair (α, β) with s stages: A = (I − α[:s])⁻¹ β[:s], b = β[s] + Aᵀ α[s]."""
        s = alpha.shape[1]
        A = np.linalg.solve(np.eye(s) - alpha[:s], beta[:s])
        b = beta[s] + A.T @ alpha[s]
        return runge_kutta_method(np.tril(A, -1), b, A.sum(1), r, **kw)

    @staticmethod
    def SSP104(**kw):  # RKMethods.jl:91-103
        s = 10
        a0 = np.diag(np.ones(s - 1), -1)
        a0[5, 4], a0[5, 0] = 2 / 5, 3 / 5
        b0 = np.diag(np.ones(s - 1), -1) / 6
        b0[5, 4] = 1 / 15
        A = np.linalg.solve(np.eye(s) - a0, b0)
        return runge_kutta_method(np.tril(A, -1), [0.1] * s, A.sum(1), 6, **kw)

    @staticmethod
    def rSSPs2(s=2, **kw):  # :106-117  (optimal low-storage s-stage 2nd order SSP)
        if s < 2:
            raise ValueError("Explicit second order SSP family requires s ≥ 2")
        r = s - 1
        alpha = np.vstack([np.zeros((1, s)), np.eye(s)])
        alpha[s, s - 1] = (s - 1) / s
        beta = alpha / r
        alpha[s, 0] = 1 / s
        return RKMethods._from_shu_osher(alpha, beta, r, **kw)

    @staticmethod
    def rSSPs3(s=4, **kw):  # :120-134  (optimal low-storage s²-stage 3rd order SSP; the reference's `s` IS the square root)
        if s < 4 or round(s**0.5) ** 2 != s:  # the reference's own check (:121-123), although it then uses s as the root: n = s²
            raise ValueError("Explicit third order SSP family requires s = n^2, n > 1")
        n = s * s
        r = n - s
        alpha = np.vstack([np.zeros((1, n)), np.eye(n)])
        alpha[s * (s + 1) // 2, s * (s + 1) // 2 - 1] = (s - 1) / (2 * s - 1)
        beta = alpha / r
        alpha[s * (s + 1) // 2, (s - 1) * (s - 2) // 2] = s / (2 * s - 1)
        return RKMethods._from_shu_osher(alpha, beta, r, **kw)

    @staticmethod
    def RK56(**kw):  # :149-162
        rows = [["1/4"], ["1/8", "1/8"], [0, 0, "1/2"], ["3/16", "-3/8", "3/8", "9/16"], ["-3/7", "8/7", "6/7", "-12/7", "8/7"]]
        return RKMethods._from_rows(rows, ["7/90", 0, "16/45", "2/15", "16/45", "7/90"], [0, "1/4", "1/4", "1/2", "3/4", 1], **kw)

    @staticmethod
    def DOPRI6(**kw):  # :165-178
        rows = [["1/5"], ["3/40", "9/40"], ["44/45", "-56/15", "32/9"], ["19372/6561", "-25360/2187", "64448/6561", "-212/729"],
                ["9017/3168", "-355/33", "46732/5247", "49/176", "-5103/18656"]]
        return RKMethods._from_rows(rows, ["35/384", 0, "500/1113", "125/192", "-2187/6784", "11/84"], **kw)

    @staticmethod
    def Mid22(**kw):  # :461-467
        return RKMethods._from_rows([["1/2"]], [0, 1], [0, "1/2"], 0.5, **kw)

    @staticmethod
    def MTE22(**kw):  # :470-476
        return RKMethods._from_rows([["2/3"]], ["1/4", "3/4"], [0, "2/3"], 0.5, **kw)

    @staticmethod
    def Heun33(**kw):  # :488-494
        return RKMethods._from_rows([["1/3"], [0, "2/3"]], ["1/4", 0, "3/4"], **kw)

    @staticmethod
    def RK33C2(**kw):  # :497-503
        return RKMethods._from_rows([["2/3"], ["1/3", "1/3"]], ["1/4", 0, "3/4"], [0, "2/3", "2/3"], **kw)

    @staticmethod
    def RK44C23(**kw):  # :533-539
        return RKMethods._from_rows([["1/2"], ["1/4", "1/4"], [0, -1, 2]], ["1/6", 0, "2/3", "1/6"], [0, "1/2", "1/2", 1], **kw)

    @staticmethod
    def RK44P2(**kw):  # :542-548
        return RKMethods._from_rows([[1], ["3/8", "1/8"], ["-1/8", "-3/8", "3/2"]], ["1/6", "-1/18", "2/3", "2/9"], [0, 1, "1/2", 1], **kw)

    @staticmethod
    def NSSP21(**kw):  # :589-598
        return RKMethods._from_rows([["3/4"]], [0, 1], [0, "3/4"], **kw)

    @staticmethod
    def NSSP32(**kw):  # :601-611
        return RKMethods._from_rows([["1/3"], [0, 1]], ["1/2", 0, "1/2"], [0, "1/3", 1], **kw)

    @staticmethod
    def NSSP33(**kw):  # :614-624
        return RKMethods._from_rows([["-4/9"], ["7/6", "-1/2"]], ["1/4", 0, "3/4"], [0, "-4/9", "2/3"], **kw)

    @staticmethod
    def NSSP53(**kw):  # :627-640
        rows = [["1/7"], [0, "3/16"], [0, 0, "1/3"], [0, 0, 0, "2/3"]]
        return RKMethods._from_rows(rows, ["1/4", 0, 0, 0, "3/4"], [0, "1/7", "3/16", "1/3", "2/3"], **kw)


class LMWray3:
    """Low memory Wray 3rd order scheme. Uses 3 vector fields and one scalar field (methods.jl:243-248)."""

    a = (8 / 15, 5 / 12, 3 / 4)
    b = (1 / 4, 0.0)
    c = (0.0, 8 / 15, 2 / 3)


def combine_(out, base, coefs, ks, setup, vector=True):
    """out = base + Σ coefs[q]·ks[q] (K6; `out` may be `base`); `vector=False`: scalar fields (the temperature)."""
    n = len(coefs)
    carr = (C.c_double * max(n, 1))(*coefs)
    karr = (C.c_void_p * max(n, 1))(*[setup.ptr(k, vector).value for k in ks])
    _lib.call("ins_combine_f64" if vector else "ins_combine_scalar_f64", setup.handle, setup.ptr(base, vector), setup.ptr(out, vector), n, carr, karr,
              setup.stream)
    return out


def _native_force(cache, setup):
    """Hand a steady body force to the native stage loop (ins_rk_set_bodyforce); the field lives in `setup`."""
    f = setup.bodyforce if (setup.bodyforce is not None and setup.issteadybodyforce) else None
    if getattr(cache, "_force", None) is not f:
        _lib.call("ins_rk_set_bodyforce", cache.handle, setup.ptr(f, True) if f is not None else None)
        cache._force = f


def _host_driven(setup, temp):
    """True when the stage loop must run on the host: a user callback or an extra equation sits between the kernels."""
    return temp is not None or setup.closure_model is not None or (setup.bodyforce is not None and not setup.issteadybodyforce) or setup.needs_bc_planes


class _TempDesc(C.Structure):
    """ins_temperature_desc_t (include/ins_hip.h)"""

    _fields_ = [("a2", C.c_double), ("a4", C.c_double), ("diss_coef", C.c_double), ("gdir", C.c_int32), ("dodissipation", C.c_int32),
                ("bc", C.c_int32 * 6), ("val", C.c_double * 6)]


def _native_ext(setup, temp, θ):
    """The native extended stage loop (csrc/ins_rk_ext.hip) serves the temperature equation and the Smagorinsky closure when nothing
    else needs the host between the kernels: returns (closure kind, θ, temperature descriptor or None), or None for the host loop."""
    if setup.needs_bc_planes or (setup.bodyforce is not None and not setup.issteadybodyforce):
        return None
    m = setup.closure_model
    kind = 0
    if m is not None:
        if getattr(m, "_ins_closure", None) != "smagorinsky" or getattr(m, "_ins_setup", None) is not setup or not np.isscalar(θ):
            return None  # a user callable: only the host can evaluate it
        kind = 1
    desc = None
    if temp is not None:
        from .operators import _temp_bc_args

        T = setup.temperature
        codes, vals, planes, _ = _temp_bc_args(setup, 0.0)
        if planes is not None:
            return None  # callable temperature boundary data
        desc = _TempDesc(a2=T.α2, a4=T.α4, diss_coef=setup.Re * T.α1 / T.γ, gdir=int(T.gdir), dodissipation=int(T.dodissipation))
        for q in range(6):
            desc.bc[q], desc.val[q] = codes[q], vals[q]
    return kind, float(θ) if kind else 0.0, desc


def _temp_rhs_(ktemp, diff, u, temp, setup):
    """ktemp = convection_diffusion_temp + dissipation (step_explicit_runge_kutta.jl:23-27)"""
    ktemp.zero_()
    convection_diffusion_temp_(ktemp, u, temp, setup)
    if setup.temperature.dodissipation:
        dissipation_(ktemp, diff, u, setup)
    return ktemp


def _lmwray3_as_erk(method):
    """The low-storage scheme IS an explicit Runge-Kutta method — its own comment spells the tableau out (step_lmwray3.jl:65-76): stage i forms
    x = xstart + Δt (Σ_{j<i} b_j k_j + a_i k_i), so in the shifted form of methods.jl:231-236 A = [[a1], [b1, a2], [b1, b2, a3]], c = (c2, c3, 1).  With
    time-independent boundary data (both of its apply_bc calls of a stage then do the same thing) the native stage loop therefore runs it: same registers
    (three vector fields: u and two stage buffers), results equal to the reference's axpy order up to rounding."""
    a, b, c = method.a, method.b, method.c
    n = len(a)
    A = np.zeros((n, n))
    for i in range(n):
        A[i, :i] = b[:i]
        A[i, i] = a[i]
    return ExplicitRungeKuttaMethod(A, A[-1].copy(), np.array(list(c[1:]) + [1.0]))


class LMWray3Cache:
    """`ode_method_cache(::LMWray3, setup)` (time_stepper_caches.jl:51-66): ustart, ONE ku, p — allocated when the host-driven loop runs (callable
    boundary data, unsteady body force, user closure model); otherwise the scheme runs inside the native stage loop as the explicit RK method it is
    (`_lmwray3_as_erk`) and `erk` holds that loop's cache."""

    def __init__(self, setup, psolver):
        self.setup, self.psolver = setup, psolver
        self._host, self._erk = None, None

    def _alloc(self):
        if self._host is None:
            setup = self.setup
            self.ustart, self.ku, self.p = vectorfield(setup), vectorfield(setup), scalarfield(setup)
            if setup.temperature is not None:
                self.tempstart, self.ktemp, self.diff = scalarfield(setup), scalarfield(setup), vectorfield(setup)
            self._host = True

    def erk(self, method):
        if self._erk is None:
            m = _lmwray3_as_erk(method)
            self._erk = (m, ERKCache(m, self.setup, self.psolver))
        return self._erk


def _timestep_lmwray3_(method, stepper, Δt, cache, θ=None):
    """step_lmwray3.jl:4-107: operator-level kernels driven from the host (with the temperature equation and the closure term)."""
    setup, psolver, u, temp, n = stepper.setup, stepper.psolver, stepper.u, stepper.temp, stepper.n
    cache._alloc()
    ustart, ku, p = cache.ustart, cache.ku, cache.p
    m = setup.closure_model
    tstart = stepper.t
    combine_(ustart, u, [], [], setup)                       # state_copyto!(xstart, x)
    if temp is not None:
        combine_(cache.tempstart, temp, [], [], setup, vector=False)
    nstage = len(method.a)
    t = tstart
    for i in range(nstage):
        t = tstart + method.c[i] * Δt
        apply_bc_u_(u, t, setup)                             # f!
        if temp is not None:
            apply_bc_temp_(temp, t, setup)
        momentum_(ku, u, temp, t, setup)
        if m is not None:
            ku.add_(m(u, θ))
        if temp is not None:
            _temp_rhs_(cache.ktemp, cache.diff, u, temp, setup)
        combine_(u, ustart, [method.a[i] * Δt], [ku], setup) # x = xstart + Δt a[i] dx
        if temp is not None:
            combine_(temp, cache.tempstart, [method.a[i] * Δt], [cache.ktemp], setup, vector=False)
        apply_bc_u_(u, t, setup)                             # correct!
        project_(u, setup, psolver, p)
        if i != nstage - 1:
            combine_(ustart, ustart, [method.b[i] * Δt], [ku], setup)
            if temp is not None:
                combine_(cache.tempstart, cache.tempstart, [method.b[i] * Δt], [cache.ktemp], setup, vector=False)
    t = tstart + Δt
    apply_bc_u_(u, t, setup)
    if temp is not None:
        apply_bc_temp_(temp, t, setup)
    return create_stepper(method, setup=setup, psolver=psolver, u=u, temp=temp, t=t, n=n + 1)


class ERKCache:
    """`ode_method_cache(method, setup)` (time_stepper_caches.jl:34-49).  The arrays live inside an
    `ins_rk_t` handle; `ustart`, `ku[i]`, `p` are exposed as zero-copy torch views where the host needs them."""

    def __init__(self, method, setup, psolver):
        self.method, self.setup, self.psolver = method, setup, psolver
        ns = len(method.b)
        A = np.ascontiguousarray(method.A, dtype=np.float64)
        c = np.ascontiguousarray(method.c, dtype=np.float64)
        self._handle = C.c_void_p()
        _lib.call("ins_rk_create", setup.handle, psolver.handle, ns, A.ctypes.data_as(_lib.c_double_p),
                  c.ctypes.data_as(_lib.c_double_p), C.byref(self._handle))

    @property
    def handle(self):
        return self._handle

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            try:
                _lib.load().ins_rk_destroy(h)
            except Exception:
                pass


def ode_method_cache(method, setup, psolver=None):
    """time_stepper_caches.jl:34-49.  (`psolver` is needed up front because the native cache owns the
    projection scratch; `solve_unsteady` passes it.)"""
    if psolver is None:
        from .pressure import default_psolver

        psolver = default_psolver(setup)
    if isinstance(method, LMWray3):
        return LMWray3Cache(setup, psolver)
    return ERKCache(method, setup, psolver)


def create_stepper(method, *, setup, psolver, u, temp=None, t=0.0, n=0):
    """step_explicit_runge_kutta.jl:1-2"""
    return SimpleNamespace(setup=setup, psolver=psolver, u=u, temp=temp, t=t, n=n)


def timesteps_(method, stepper, Δt, nsteps, *, θ=None, cache):
    """`nsteps` time steps of size Δt, in place: the fixed-Δt loop of solve_unsteady (solver.jl:74-83) as one native call
    (`ins_rk_steps_f64`) when the boundary data is time-independent; otherwise `nsteps` calls of `timestep_`.
    The stepper's `u` is valid on entry and on return (intermediate steps are not observable, as inside the reference's loop
    without processors)."""
    setup, psolver = stepper.setup, stepper.psolver
    if nsteps >= 1 and isinstance(method, LMWray3) and not _host_driven(setup, stepper.temp) and not os.environ.get("INS_HOST_STAGE_LOOP"):
        m_erk, c_erk = cache.erk(method)
        st = timesteps_(m_erk, stepper, Δt, nsteps, θ=θ, cache=c_erk)
        return create_stepper(method, setup=setup, psolver=psolver, u=st.u, temp=None, t=stepper.t + nsteps * Δt, n=stepper.n + nsteps)
    if nsteps >= 1 and not isinstance(method, LMWray3) and not _host_driven(setup, stepper.temp):
        if cache.psolver is not psolver:
            raise ValueError("cache was created for a different psolver")
        _native_force(cache, setup)
        _lib.call("ins_rk_steps_f64", cache.handle, 1.0 / setup.Re, setup.ptr(stepper.u, True), float(stepper.t), float(Δt), int(nsteps), setup.stream)
        return create_stepper(method, setup=setup, psolver=psolver, u=stepper.u, temp=None, t=stepper.t + nsteps * method.c[-1] * Δt, n=stepper.n + nsteps)
    for _ in range(nsteps):
        stepper = timestep_(method, stepper, Δt, θ=θ, cache=cache)
    return stepper


def timestep_(method, stepper, Δt, *, θ=None, cache):
    """Perform one time step, in place (step_explicit_runge_kutta.jl:4-59).

    Fully native (`ins_rk_step_f64`) when boundary data is time-independent; with callable Dirichlet data
    the stage loop runs here and calls the operator-level kernels so `bc.u(α, x..., t)` can be evaluated
    between stages (SURVEY.md §8b closure-hook caveat)."""
    setup, psolver, u, temp, t, n = stepper.setup, stepper.psolver, stepper.u, stepper.temp, stepper.t, stepper.n
    if (temp is None) != (setup.temperature is None):
        raise ValueError("a temperature field needs setup.temperature (temperature_equation) and vice versa")
    if isinstance(method, LMWray3):
        # native when nothing has to run on the host between the kernels: the scheme as the explicit RK method it is, inside the native stage loop
        if os.environ.get("INS_HOST_STAGE_LOOP") or (_host_driven(setup, temp) and _native_ext(setup, temp, θ) is None):
            return _timestep_lmwray3_(method, stepper, Δt, cache, θ)
        m_erk, c_erk = cache.erk(method)
        st = timestep_(m_erk, stepper, Δt, θ=θ, cache=c_erk)
        return create_stepper(method, setup=setup, psolver=psolver, u=st.u, temp=st.temp, t=t + Δt, n=n + 1)
    if cache.psolver is not psolver:
        raise ValueError("cache was created for a different psolver")
    if not _host_driven(setup, temp):
        _native_force(cache, setup)
        _lib.call("ins_rk_step_f64", cache.handle, 1.0 / setup.Re, setup.ptr(u, True), float(t), float(Δt), None, setup.stream)
        return create_stepper(method, setup=setup, psolver=psolver, u=u, temp=None, t=stepper.t + method.c[-1] * Δt, n=n + 1)
    if (setup.needs_bc_planes and temp is None and setup.closure_model is None and (setup.bodyforce is None or setup.issteadybodyforce)
            and not os.environ.get("INS_HOST_STAGE_LOOP")):
        # callable Dirichlet data is the only thing that needs the host: the ghost fills of the stage loop happen at times known before the step
        # (tstart and tstart + c[i] Δt), so the closures are evaluated for those now and the whole stage loop runs natively (ins_rk_step_bc_f64)
        from .operators import _bc_planes

        ns = len(method.b)
        sets = (C.c_void_p * (18 * (ns + 1)))()
        keep = []
        for q in range(ns + 1):
            arr, k = _bc_planes(setup, t if q == 0 else t + method.c[q - 1] * Δt, False)
            keep.append(k)
            for j in range(18):
                sets[18 * q + j] = arr[j]
        _native_force(cache, setup)
        _lib.call("ins_rk_step_bc_f64", cache.handle, 1.0 / setup.Re, setup.ptr(u, True), float(t), float(Δt), sets, setup.stream)
        return create_stepper(method, setup=setup, psolver=psolver, u=u, temp=None, t=stepper.t + method.c[-1] * Δt, n=n + 1)
    ext = None if os.environ.get("INS_HOST_STAGE_LOOP") else _native_ext(setup, temp, θ)
    if ext is not None:  # temperature equation / Smagorinsky closure inside the native loop
        kind, th, desc = ext
        _native_force(cache, setup)
        _lib.call("ins_rk_set_closure", cache.handle, kind, th)
        _lib.call("ins_rk_set_temperature", cache.handle, C.byref(desc) if desc is not None else None)
        _lib.call("ins_rk_step_ext_f64", cache.handle, 1.0 / setup.Re, setup.ptr(u, True), setup.ptr(temp, False) if temp is not None else None,
                  float(t), float(Δt), setup.stream)
        return create_stepper(method, setup=setup, psolver=psolver, u=u, temp=temp, t=stepper.t + method.c[-1] * Δt, n=n + 1)
    # host-driven stage loop (time-dependent boundary data, unsteady body force, user closure model)
    A, c = method.A, method.c
    ns = len(method.b)
    m = setup.closure_model
    if not hasattr(cache, "_host"):
        cache._host = dict(ustart=vectorfield(setup), ku=[vectorfield(setup) for _ in range(ns)], p=scalarfield(setup))
        if setup.temperature is not None:  # time_stepper_caches.jl:40-47
            cache._host.update(tempstart=scalarfield(setup), ktemp=[scalarfield(setup) for _ in range(ns)], diff=vectorfield(setup))
    H = cache._host
    ustart, ku, p = H["ustart"], H["ku"], H["p"]
    tstart = t
    combine_(ustart, u, [], [], setup)
    if temp is not None:
        combine_(H["tempstart"], temp, [], [], setup, vector=False)
    for i in range(ns):
        apply_bc_u_(u, t, setup)
        if temp is not None:
            apply_bc_temp_(temp, t, setup)
        momentum_(ku[i], u, temp, t, setup)
        if temp is not None:
            _temp_rhs_(H["ktemp"][i], H["diff"], u, temp, setup)
        if m is not None:
            ku[i].add_(m(u, θ))
        t = tstart + c[i] * Δt
        combine_(u, ustart, [Δt * A[i, j] for j in range(i + 1)], [ku[j] for j in range(i + 1)], setup)
        if temp is not None:
            combine_(temp, H["tempstart"], [Δt * A[i, j] for j in range(i + 1)], [H["ktemp"][j] for j in range(i + 1)], setup, vector=False)
        apply_bc_u_(u, t, setup)
        project_(u, setup, psolver, p)
    apply_bc_u_(u, t, setup)
    if temp is not None:
        apply_bc_temp_(temp, t, setup)
    return create_stepper(method, setup=setup, psolver=psolver, u=u, temp=temp, t=t, n=n + 1)


def timestep(method, stepper, Δt, *, θ=None):
    """Out-of-place twin (step_explicit_runge_kutta.jl:61-120): same result on a copy of `u`."""
    cache = ode_method_cache(method, stepper.setup, stepper.psolver)
    s2 = create_stepper(method, setup=stepper.setup, psolver=stepper.psolver, u=copyfield(stepper.u),
                        temp=None if stepper.temp is None else copyfield(stepper.temp), t=stepper.t, n=stepper.n)
    return timestep_(method, s2, Δt, θ=θ, cache=cache)
