"""Operator API of operators.jl / boundary_conditions.jl: in-place `op_(out, in, setup)` (Julia's `op!`)
and allocating `op(in, setup)` twins, each a thin call into libinship's HIP kernels."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .boundary_conditions import DirichletBC
from .setup import copyfield, scalarfield, vectorfield


# ------------------------------------------------------------------------------------ ghost fill
def _bc_planes(setup, t, dudt):
    """Evaluate callable Dirichlet data `bc.u(alpha, x..., t)` on the boundary planes
    (boundary_conditions.jl:351-372) into device plane buffers; returns (ctypes array | None, keepalive)."""
    if not setup.needs_bc_planes:
        return None, None
    g = setup.grid
    D = g.dimension
    arr = (C.c_void_p * 18)()
    keep = []
    for be in range(D):
        others = [b for b in range(D) if b != be]
        for side in range(2):
            bc = setup.boundary_conditions[be][side]
            if not (isinstance(bc, DirichletBC) and callable(bc.u)):
                continue
            for al in range(D):
                lo, hi = g.Iu[al][be]
                i = hi if side else lo - 1
                xs = []
                for ga in range(D):
                    coords = g.xu[al][ga][i : i + 1] if ga == be else g.xu[al][ga]
                    shape = [1] * D
                    shape[ga] = coords.size
                    xs.append(coords.reshape(shape))
                if dudt:
                    h = np.sqrt(np.finfo(np.float64).eps) / 2
                    val = (bc.u(al, *xs, t + h) - bc.u(al, *xs, t - h)) / (2 * h)
                else:
                    val = bc.u(al, *xs, t)
                full = np.broadcast_to(val, [1 if b == be else g.N[b] for b in range(D)])
                plane = np.ascontiguousarray(np.transpose(np.squeeze(full, axis=be)))  # memory order: fastest dir first
                buf = torch.from_numpy(plane.astype(np.float64)).to(setup.device)
                keep.append(buf)
                arr[(be * 2 + side) * 3 + al] = buf.data_ptr()
    return arr, keep


def apply_bc_u_(u, t, setup, dudt=False):
    """Apply velocity boundary conditions, in place (boundary_conditions.jl:159-167)."""
    planes, keep = _bc_planes(setup, t, dudt)
    _lib.call("ins_apply_bc_u_f64", setup.handle, setup.ptr(u, True), int(dudt), planes, setup.stream)
    return u


def apply_bc_p_(p, t, setup):
    """Apply pressure boundary conditions, in place (boundary_conditions.jl:197-206)."""
    _lib.call("ins_apply_bc_p_f64", setup.handle, setup.ptr(p, False), setup.stream)
    return p


def apply_bc_u(u, t, setup, **kw):
    return apply_bc_u_(copyfield(u), t, setup, **kw)


def apply_bc_p(p, t, setup):
    return apply_bc_p_(copyfield(p), t, setup)


# ------------------------------------------------------------------------------------ operators.jl
def scalewithvolume_(p, setup):
    """operators.jl:81-95"""
    _lib.call("ins_scalewithvolume_f64", setup.handle, setup.ptr(p, False), setup.stream)
    return p


def scalewithvolume(p, setup):
    return scalewithvolume_(copyfield(p), setup)


def divergence_(div, u, setup):
    """operators.jl:106-115"""
    _lib.call("ins_divergence_f64", setup.handle, setup.ptr(u, True), setup.ptr(div, False), setup.stream)
    return div


def divergence(u, setup):
    return divergence_(scalarfield(setup), u, setup)


def pressuregradient_(G, p, setup):
    """operators.jl:159-168"""
    _lib.call("ins_pressuregradient_f64", setup.handle, setup.ptr(p, False), setup.ptr(G, True), setup.stream)
    return G


def pressuregradient(p, setup):
    return pressuregradient_(vectorfield(setup), p, setup)


def applypressure_(u, p, setup):
    """operators.jl:214-223"""
    _lib.call("ins_applypressure_f64", setup.handle, setup.ptr(u, True), setup.ptr(p, False), setup.stream)
    return u


def applypressure(u, p, setup):
    return applypressure_(copyfield(u), p, setup)


def laplacian_(L, p, setup):
    """operators.jl:297-364"""
    _lib.call("ins_laplacian_f64", setup.handle, setup.ptr(p, False), setup.ptr(L, False), setup.stream)
    return L


def laplacian(p, setup):
    return laplacian_(scalarfield(setup), p, setup)


def convection_(F, u, setup):
    """operators.jl:378-387 (adds to F)"""
    _lib.call("ins_convection_f64", setup.handle, setup.ptr(u, True), setup.ptr(F, True), setup.stream)
    return F


def convection(u, setup):
    return convection_(vectorfield(setup), u, setup)


def diffusion_(F, u, setup, use_viscosity=True):
    """operators.jl:537-547 (adds to F)"""
    visc = 1.0 / setup.Re if use_viscosity else 1.0
    _lib.call("ins_diffusion_f64", setup.handle, visc, setup.ptr(u, True), setup.ptr(F, True), setup.stream)
    return F


def diffusion(u, setup, use_viscosity=True):
    return diffusion_(vectorfield(setup), u, setup, use_viscosity)


def convectiondiffusion_(F, u, setup):
    """operators.jl:634-645 (adds to F)"""
    _lib.call("ins_convectiondiffusion_f64", setup.handle, 1.0 / setup.Re, setup.ptr(u, True), setup.ptr(F, True), setup.stream)
    return F


def momentum_(F, u, temp, t, setup):
    """operators.jl:967-976: fill!(F, 0) + convectiondiffusion! (one fused write-only pass), then the body force and gravity terms."""
    _lib.call("ins_momentum_f64", setup.handle, 1.0 / setup.Re, setup.ptr(u, True), setup.ptr(F, True), setup.stream)
    if setup.bodyforce is not None:
        applybodyforce_(F, u, t, setup)
    if temp is not None:
        gravity_(F, temp, setup)
    return F


def momentum(u, temp, t, setup):
    return momentum_(vectorfield(setup), u, temp, t, setup)


def kinetic_energy_(ke, u, setup, interpolate_first=False):
    """operators.jl:1516-1545"""
    _lib.call("ins_kinetic_energy_f64", setup.handle, setup.ptr(u, True), setup.ptr(ke, False), int(interpolate_first), setup.stream)
    return ke


def kinetic_energy(u, setup, **kw):
    return kinetic_energy_(scalarfield(setup), u, setup, **kw)


def total_kinetic_energy(u, setup, interpolate_first=False):
    """operators.jl:1551-1556 (blocking)"""
    out = C.c_double()
    _lib.call("ins_total_kinetic_energy_f64", setup.handle, setup.ptr(u, True), int(interpolate_first), C.byref(out), setup.stream)
    return out.value


def max_abs_divergence(u, setup):
    """maximum(abs, divergence(u, setup)[Ip]) (blocking)"""
    out = C.c_double()
    _lib.call("ins_max_abs_divergence_f64", setup.handle, setup.ptr(u, True), C.byref(out), setup.stream)
    return out.value


# ------------------------------------------------------------------------------------ body force
def applybodyforce_(F, u, t, setup):
    """operators.jl:873-897 (adds to F): the stored field when steady, else `bodyforce.(α, xu[α]..., t)` evaluated on the host."""
    f = setup.bodyforce if setup.issteadybodyforce else setup.bodyforce_field(t)
    setup.ptr(F, True), setup.ptr(f, True)
    F.add_(f)  # same-layout elementwise add (torch: plumbing, as the reference's broadcast)
    return F


def applybodyforce(u, t, setup):
    """operators.jl:856-866"""
    return copyfield(setup.bodyforce) if setup.issteadybodyforce else setup.bodyforce_field(t)


# ------------------------------------------------------------------------------------ temperature equation
def _temp_bc_args(setup, t):
    g = setup.grid
    D = g.dimension
    bcs = setup.temperature.boundary_conditions
    codes = (C.c_int32 * 6)()
    vals = (C.c_double * 6)()
    planes = (C.c_void_p * 6)()
    keep, anyplane = [], False
    for be in range(D):
        for side in range(2):
            bc = bcs[be][side]
            codes[2 * be + side] = bc.code
            if not isinstance(bc, DirichletBC) or bc.u is None:
                continue
            if callable(bc.u):  # bc.u(x..., t) on the full padded plane (boundary_conditions.jl:391-405)
                lo, hi = g.Ip[be]
                i = hi if side else lo - 1
                xs = []
                for ga in range(D):
                    coords = g.xp[ga][i : i + 1] if ga == be else g.xp[ga]
                    shape = [1] * D
                    shape[ga] = coords.size
                    xs.append(np.asarray(coords).reshape(shape))
                full = np.broadcast_to(bc.u(*xs, t), [1 if b == be else g.N[b] for b in range(D)])
                plane = np.ascontiguousarray(np.transpose(np.squeeze(full, axis=be)))  # memory order: fastest direction first
                buf = torch.from_numpy(plane.astype(np.float64)).to(setup.device)
                keep.append(buf)
                planes[2 * be + side] = buf.data_ptr()
                anyplane = True
            else:
                vals[2 * be + side] = float(bc.u)
    return codes, vals, (planes if anyplane else None), keep


def apply_bc_temp_(temp, t, setup):
    """Apply temperature boundary conditions, in place (boundary_conditions.jl:236-246)."""
    codes, vals, planes, keep = _temp_bc_args(setup, t)
    _lib.call("ins_apply_bc_temp_f64", setup.handle, codes, vals, planes, setup.ptr(temp, False), setup.stream)
    if keep:
        torch.cuda.current_stream(setup.device).synchronize()  # the plane buffers die with this frame
    return temp


def apply_bc_temp(temp, t, setup):
    return apply_bc_temp_(copyfield(temp), t, setup)


def convection_diffusion_temp_(c, u, temp, setup):
    """operators.jl:712-737 (adds to c)"""
    _lib.call("ins_convection_diffusion_temp_f64", setup.handle, setup.temperature.α4, setup.ptr(u, True), setup.ptr(temp, False),
              setup.ptr(c, False), setup.stream)
    return c


def convection_diffusion_temp(u, temp, setup):
    return convection_diffusion_temp_(scalarfield(setup), u, temp, setup)


def dissipation_(diss, diff, u, setup):
    """operators.jl:791-814 (adds to diss; `diff` is scratch: it receives diffusion(u))"""
    T = setup.temperature
    _lib.call("ins_dissipation_f64", setup.handle, 1.0 / setup.Re, setup.Re * T.α1 / T.γ, setup.ptr(u, True), setup.ptr(diff, True),
              setup.ptr(diss, False), setup.stream)
    return diss


def dissipation(u, setup):
    return dissipation_(scalarfield(setup), vectorfield(setup), u, setup)


def dissipation_from_strain_(ϵ, u, setup):
    """operators.jl:836-854"""
    _lib.call("ins_dissipation_from_strain_f64", setup.handle, 1.0 / setup.Re, setup.ptr(u, True), setup.ptr(ϵ, False), setup.stream)
    return ϵ


def dissipation_from_strain(u, setup):
    return dissipation_from_strain_(scalarfield(setup), u, setup)


def gravity_(F, temp, setup):
    """operators.jl:914-931 (adds to F)"""
    T = setup.temperature
    _lib.call("ins_gravity_f64", setup.handle, int(T.gdir), T.α2, setup.ptr(temp, False), setup.ptr(F, True), setup.stream)
    return F


def gravity(temp, setup):
    return gravity_(vectorfield(setup), temp, setup)


# ------------------------------------------------------------------------------------ field diagnostics
def _vort_field(setup):
    return scalarfield(setup) if setup.grid.dimension == 2 else vectorfield(setup)


def vorticity_(ω, u, setup):
    """operators.jl:985-1020"""
    _lib.call("ins_vorticity_f64", setup.handle, setup.ptr(u, True), setup.ptr(ω, setup.grid.dimension == 3), setup.stream)
    return ω


def vorticity(u, setup):
    return vorticity_(_vort_field(setup), u, setup)


def interpolate_u_p_(up, u, setup):
    """operators.jl:1311-1326"""
    _lib.call("ins_interpolate_u_p_f64", setup.handle, setup.ptr(u, True), setup.ptr(up, True), setup.stream)
    return up


def interpolate_u_p(u, setup):
    return interpolate_u_p_(vectorfield(setup), u, setup)


def interpolate_ω_p_(ωp, ω, setup):
    """operators.jl:1336-1370"""
    vec = setup.grid.dimension == 3
    _lib.call("ins_interpolate_w_p_f64", setup.handle, setup.ptr(ω, vec), setup.ptr(ωp, vec), setup.stream)
    return ωp


def interpolate_ω_p(ω, setup):
    return interpolate_ω_p_(_vort_field(setup), ω, setup)


def Dfield_(d, G, p, setup, ϵ=np.finfo(np.float64).eps):
    """operators.jl:1385-1422"""
    _lib.call("ins_dfield_f64", setup.handle, setup.ptr(p, False), setup.ptr(G, True), setup.ptr(d, False), float(ϵ), setup.stream)
    return d


def Dfield(p, setup, **kw):
    return Dfield_(scalarfield(setup), vectorfield(setup), p, setup, **kw)


def Qfield_(Q, u, setup):
    """operators.jl:1440-1460"""
    _lib.call("ins_qfield_f64", setup.handle, setup.ptr(u, True), setup.ptr(Q, False), setup.stream)
    return Q


def Qfield(u, setup):
    return Qfield_(scalarfield(setup), u, setup)


def eig2field_(λ, u, setup):
    """operators.jl:1472-1492 (3-D only)"""
    _lib.call("ins_eig2field_f64", setup.handle, setup.ptr(u, True), setup.ptr(λ, False), setup.stream)
    return λ


def eig2field(u, setup):
    return eig2field_(scalarfield(setup), u, setup)


# ------------------------------------------------------------------------------------ Smagorinsky closure
def tensorfield(setup):
    """Symmetric tensor field: D(D+1)/2 scalar fields [xx, yy, (zz), xy, (xz, yz)] (the reference stores a D×D SMatrix per cell)."""
    from .setup import _alloc

    D = setup.grid.dimension
    return _alloc(setup, setup.grid.N + (D * (D + 1) // 2,))


def smagtensor_(σ, u, θ, setup):
    """operators.jl:1135-1150"""
    D = setup.grid.dimension
    _lib.call("ins_smagtensor_f64", setup.handle, float(θ), setup.ptr(u, True), setup.ptr(σ, D * (D + 1) // 2), setup.stream)
    return σ


def divoftensor_(s, σ, setup):
    """operators.jl:1158-1175, 1203-1236"""
    D = setup.grid.dimension
    _lib.call("ins_divoftensor_f64", setup.handle, setup.ptr(σ, D * (D + 1) // 2), setup.ptr(s, True), setup.stream)
    return s


def smagorinsky_closure(setup):
    """Create Smagorinsky closure model `m(u, θ)` (operators.jl:1284-1300)."""
    s = vectorfield(setup)
    D = setup.grid.dimension
    scratch = {}

    def closure(u, θ):
        # smagtensor! -> apply_bc_p!(σ) -> divoftensor! behind one entry point: a single kernel wherever the grid allows it (csrc/ins_smagforce.hip:
        # the stress stays in registers, no σ field is allocated), the three kernels with σ as scratch elsewhere — asked per call, an option may
        # switch the route
        σp = None
        if _lib.load().ins_smagorinsky_force_needs_sigma(setup.handle):
            if "σ" not in scratch:
                scratch["σ"] = tensorfield(setup)
            σp = setup.ptr(scratch["σ"], D * (D + 1) // 2)
        _lib.call("ins_smagorinsky_force_f64", setup.handle, float(θ), setup.ptr(u, True), σp, setup.ptr(s, True), setup.stream)
        return s

    closure._ins_closure, closure._ins_setup = "smagorinsky", setup  # lets timestep_ run it inside the native stage loop (ins_rk_set_closure)
    return closure


# ------------------------------------------------------------------------------------ tensor basis
def tensorbasis_(B, V, u, setup):
    """tensorbasis.jl:16-28.  `B`: N + (nb·D·D,) with element (a, b) of tensor ib at ib·D·D + a + D·b; `V`: N + (nv,)."""
    D = setup.grid.dimension
    nb, nv = (3, 2) if D == 2 else (11, 5)
    _lib.call("ins_tensorbasis_f64", setup.handle, setup.ptr(u, True), setup.ptr(B, nb * D * D), setup.ptr(V, nv), setup.stream)
    return B, V


def tensorbasis(u, setup):
    """Compute symmetry tensor basis B[1]..B[nb] and invariants V[1]..V[nv] (tensorbasis.jl:1-10): returns `(B, V)` with
    `B.reshape(N + (nb, D, D))[..., ib, b, a]` = element (a, b) of tensor ib (use `tensorbasis_matrices` for that view)."""
    from .setup import _alloc

    D = setup.grid.dimension
    nb, nv = (3, 2) if D == 2 else (11, 5)
    return tensorbasis_(_alloc(setup, setup.grid.N + (nb * D * D,)), _alloc(setup, setup.grid.N + (nv,)), u, setup)


def tensorbasis_matrices(B, setup):
    """View of `B` as N + (nb, D, D) with [..., ib, a, b] = element (a, b)."""
    D = setup.grid.dimension
    return B.reshape(tuple(setup.grid.N) + (-1, D, D)).transpose(-1, -2)
