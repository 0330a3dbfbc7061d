"""Boundary-condition types and ghost-layout rules (boundary_conditions.jl:2-103), host side."""
from . import _lib


class AbstractBC:
    code = None

    def __repr__(self):
        return f"{type(self).__name__}()"


class PeriodicBC(AbstractBC):
    """Periodic boundary conditions. Must be periodic on both sides. (boundary_conditions.jl:4-5)"""

    code = _lib.INS_BC_PERIODIC


class DirichletBC(AbstractBC):
    """Dirichlet BC for the velocity (boundary_conditions.jl:7-19): `u` is None (no slip), a tuple of
    constants, or a callable `(alpha, x..., t)` (alpha 0-based here)."""

    code = _lib.INS_BC_DIRICHLET

    def __init__(self, u=None):
        self.u = u

    def __repr__(self):
        return f"DirichletBC({self.u!r})"


class SymmetricBC(AbstractBC):
    """boundary_conditions.jl:21-26"""

    code = _lib.INS_BC_SYMMETRIC


class PressureBC(AbstractBC):
    """boundary_conditions.jl:28-36"""

    code = _lib.INS_BC_PRESSURE


class HaloBC(AbstractBC):
    """Not a physical boundary: the z-face of a slab whose ghost plane is filled by the neighbouring rank
    (multi-GPU decomposition).  Ghost layout as PeriodicBC (one ghost volume, the width of the
    adjacent interior volume — exact on the uniform grids the spectral solver requires)."""

    code = _lib.INS_BC_HALO


def padghost_(bc, x, isright):
    """boundary_conditions.jl:42-61 on a python list."""
    if isinstance(bc, PeriodicBC):
        x.append(x[-1] + (x[1] - x[0])) if isright else x.insert(0, x[0] - (x[-1] - x[-2]))
    elif isinstance(bc, DirichletBC):
        x.append(x[-1]) if isright else x.insert(0, x[0])
    elif isinstance(bc, (SymmetricBC, HaloBC)):
        x.append(x[-1] + (x[-1] - x[-2])) if isright else x.insert(0, x[0] - (x[1] - x[0]))
    elif isinstance(bc, PressureBC):
        if isright:
            x.append(x[-1])
        else:
            x.insert(0, x[0])
            x.insert(0, x[0])
    else:
        raise TypeError(f"not a boundary condition: {bc!r}")


def offset_u(bc, isright, isnormal):
    """boundary_conditions.jl:79-88"""
    if isinstance(bc, (PeriodicBC, HaloBC)):
        return 1
    if isinstance(bc, (DirichletBC, SymmetricBC)):
        return 1 + int(isright and isnormal)
    if isinstance(bc, PressureBC):
        return 1 + int((not isright) and (not isnormal))
    raise TypeError(bc)


def offset_p(bc, isright):
    """boundary_conditions.jl:80-89"""
    if isinstance(bc, PressureBC):
        return 1 + int(not isright)
    if isinstance(bc, AbstractBC):
        return 1
    raise TypeError(bc)
