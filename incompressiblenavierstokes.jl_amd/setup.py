"""`Setup` (setup.jl:2-46) and the field allocators (initializers.jl:2-6), host side.

Fields are torch tensors (device memory only) with the reference's memory layout: a vector field has
shape `N + (D,)` and Fortran strides, so `u[i, j, k, a]` is Julia's `u[i+1, j+1, k+1, a+1]` and
`u.data_ptr()` can be handed to libinship (or to a Julia `unsafe_wrap`) unchanged.
"""
import ctypes as C
import math
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib
from .boundary_conditions import DirichletBC, PeriodicBC
from .grid import Grid


def _fortran_strides(shape):
    st, acc = [], 1
    for n in shape:
        st.append(acc)
        acc *= n
    return tuple(st)


class Temperature(SimpleNamespace):
    """The named tuple `temperature_equation` returns (setup.jl:48-87): α1..α4, γ, dodissipation, boundary_conditions, gdir (0-based)."""


def temperature_equation(*, Pr, Ra, Ge, boundary_conditions, dodissipation=True, gdir=1, nondim_type=1):
    """Create temperature equation setup (setup.jl:48-87).  `gdir` is 0-based here (the reference's default 2 is 1)."""
    if nondim_type == 1:  # free fall velocity
        a1, a2, a3, a4 = math.sqrt(Pr / Ra), 1.0, Ge * math.sqrt(Pr / Ra), 1 / math.sqrt(Pr * Ra)
    elif nondim_type == 2:  # heat conduction time scale
        a1, a2, a3, a4 = Pr, Pr * Ra, Ge / Ra, 1.0
    elif nondim_type == 3:
        a1, a2, a3, a4 = math.sqrt(Pr * Ge / Ra), Ge, math.sqrt(Pr * Ge / Ra), math.sqrt(Ge / (Pr * Ra))
    else:
        raise ValueError(f"nondim_type = {nondim_type}")
    bcs = tuple(tuple(side) for side in boundary_conditions)
    return Temperature(α1=a1, α2=a2, α3=a3, α4=a4, γ=a1 / a3, dodissipation=bool(dodissipation), boundary_conditions=bcs, gdir=int(gdir))


class Setup:
    """Problem setup (setup.jl:2-46).  `backend`/`workgroupsize` of the reference are replaced by
    `device` (a torch CUDA/HIP device); everything else keeps its name.

    `bodyforce(α, x..., t)` (α 0-based) is evaluated on the host over the padded coordinates `xu[α]` and uploaded: once, at t = 0,
    when `issteadybodyforce` (setup.jl:25-32), otherwise at every `applybodyforce_` call.  `closure_model(u, θ)` returns a vector
    field on the device (e.g. `smagorinsky_closure(setup)`).  With any of the three the time steppers drive the operator-level
    kernels from the host (closure hook caveat); without them a step is one native call."""

    def __init__(self, *, x, boundary_conditions=None, Re=None, bodyforce=None, issteadybodyforce=True, closure_model=None,
                 temperature=None, device=None):
        D = len(x)
        if boundary_conditions is None:
            boundary_conditions = tuple((PeriodicBC(), PeriodicBC()) for _ in range(D))
        self.boundary_conditions = tuple(tuple(side) for side in boundary_conditions)
        for a, b in self.boundary_conditions:
            if isinstance(a, PeriodicBC) != isinstance(b, PeriodicBC):
                raise ValueError("PeriodicBC must be periodic on both sides")
        self.grid = Grid(x, self.boundary_conditions)
        if Re is None:  # setup.jl:12
            Re = 1000.0 if temperature is None else 1.0 / temperature.α1
        self.Re = float(Re)
        self.bodyforce = None
        self.issteadybodyforce = False
        self.closure_model = closure_model
        self.temperature = temperature
        if temperature is not None:
            if len(temperature.boundary_conditions) != D:
                raise ValueError("temperature boundary conditions: one (left, right) pair per direction")
            for a, b in temperature.boundary_conditions:
                if isinstance(a, PeriodicBC) != isinstance(b, PeriodicBC):
                    raise ValueError("PeriodicBC must be periodic on both sides")
        if device is None:
            if not torch.cuda.is_available():
                raise _lib.INSHipError("no HIP device visible: the accelerated path has no CPU fallback")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.INSHipError(f"Setup needs a HIP device, got {self.device}")
        self._handle = None
        self._make_handle()
        if bodyforce is not None:
            if not callable(bodyforce):
                raise TypeError("bodyforce must be callable: bodyforce(α, x..., t)")
            self.bodyforce = bodyforce
            if issteadybodyforce:  # setup.jl:25-32
                self.bodyforce = self.bodyforce_field(0.0)
                self.issteadybodyforce = True

    def bodyforce_field(self, t):
        """`bodyforce.(α, xu[α]..., t)` over the whole padded arrays (operators.jl:885-896), as a device vector field."""
        g = self.grid
        D = g.dimension
        F = np.zeros(g.N + (D,))
        for a in range(D):
            xs = []
            for b in range(D):
                shape = [1] * D
                shape[b] = g.N[b]
                xs.append(np.asarray(g.xu[a][b]).reshape(shape))
            F[..., a] = np.broadcast_to(self.bodyforce(a, *xs, t), g.N)
        return from_numpy(self, F)

    # -------------------------------------------------------------------------------------------
    def _desc(self):
        g = self.grid
        D = g.dimension
        d = _lib.GridDesc()
        d.D = D
        keep = []

        def ptr(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            keep.append(a)
            return a.ctypes.data_as(_lib.c_double_p)

        for a in range(D):
            d.N[a] = g.N[a]
            d.dx[a] = ptr(g.Δ[a])
            d.dxu[a] = ptr(g.Δu[a])
            d.ip_lo[a], d.ip_hi[a] = g.Ip[a]
            for b in range(D):
                d.A1[a][b] = ptr(g.A[a][b][0])
                d.A2[a][b] = ptr(g.A[a][b][1])
                d.iu_lo[a][b], d.iu_hi[a][b] = g.Iu[a][b]
            for side in range(2):
                bc = self.boundary_conditions[a][side]
                d.bc[a][side] = bc.code
                if isinstance(bc, DirichletBC) and isinstance(bc.u, tuple):
                    for c in range(D):
                        d.bc_u[a][side][c] = float(bc.u[c])
        return d, keep

    def _make_handle(self):
        _lib.call("ins_set_device", self.device.index or 0)
        d, keep = self._desc()
        h = C.c_void_p()
        _lib.call("ins_grid_create", C.byref(d), C.byref(h))
        self._handle = h
        del keep

    @property
    def handle(self):
        return self._handle

    @property
    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def needs_bc_planes(self):
        return any(isinstance(bc, DirichletBC) and callable(bc.u) for side in self.boundary_conditions for bc in side)

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            try:
                _lib.load().ins_grid_destroy(h)
            except Exception:
                pass

    # -------------------------------------------------------------------------------------------
    def ptr(self, f, vector):
        """Device pointer of a field after checking dtype / device / shape / reference layout.
        `vector`: False (scalar field), True (D components) or an int (that many components, e.g. a symmetric tensor)."""
        g = self.grid
        ncomp = g.dimension if vector is True else int(vector)
        shape = g.N + ((ncomp,) if ncomp else ())
        if not isinstance(f, torch.Tensor) or f.dtype != torch.float64:
            raise TypeError("fields must be float64 torch tensors")
        if f.device != self.device:
            raise ValueError(f"field lives on {f.device}, setup on {self.device}")
        if tuple(f.shape) != shape or tuple(f.stride()) != _fortran_strides(shape):
            raise ValueError(f"field must have shape {shape} with column-major strides (use scalarfield/vectorfield)")
        return C.c_void_p(f.data_ptr())


def _alloc(setup, shape):
    t = torch.zeros(tuple(reversed(shape)), dtype=torch.float64, device=setup.device)
    return t.permute(*reversed(range(len(shape))))


def scalarfield(setup):
    """Create empty scalar field (initializers.jl:2)."""
    return _alloc(setup, setup.grid.N)


def vectorfield(setup):
    """Create empty vector field (initializers.jl:5-6)."""
    return _alloc(setup, setup.grid.N + (setup.grid.dimension,))


def from_numpy(setup, a):
    """Upload a host array of field shape into a fresh device field (keeps the reference layout)."""
    a = np.asarray(a, dtype=np.float64)
    f = _alloc(setup, a.shape)
    f.copy_(torch.from_numpy(np.ascontiguousarray(a)).to(setup.device))
    return f


def to_numpy(f):
    """Download a field as a Fortran-ordered numpy array of the same shape."""
    return np.asfortranarray(f.detach().cpu().numpy())


def copyfield(f):
    """`copy(u)` preserving the column-major layout."""
    out = torch.empty(tuple(reversed(f.shape)), dtype=f.dtype, device=f.device).permute(*reversed(range(f.dim())))
    out.copy_(f)
    return out
