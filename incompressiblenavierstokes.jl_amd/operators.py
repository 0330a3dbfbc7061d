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
    """operators.jl:967-976 with bodyforce = temp = nothing"""
    if temp is not None:
        raise NotImplementedError("temperature is outside the HIP hot path")
    _lib.call("ins_momentum_f64", setup.handle, 1.0 / setup.Re, setup.ptr(u, True), setup.ptr(F, True), setup.stream)
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
