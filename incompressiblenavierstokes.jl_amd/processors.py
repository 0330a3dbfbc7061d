"""Processors (processors.jl): callbacks that observe the stepper state during `solve_unsteady`, and the on-device observers
they are built from.  The reference's `Observable` is a minimal value-with-listeners object here."""
import base64
import ctypes as C
import math
import os
import time

import numpy as np
import torch

from . import _lib
from .operators import (Dfield_, Qfield_, apply_bc_u_, eig2field_, interpolate_u_p_, interpolate_ω_p_, vorticity_)
from .pressure import default_psolver, pressure
from .setup import scalarfield, to_numpy, vectorfield


class Observable:
    """Value with listeners (Observables.jl as used by processors.jl)."""

    def __init__(self, value):
        self._value, self._listeners = value, []

    @property
    def value(self):
        return self._value

    @value.setter
    def value(self, v):
        self._value = v
        for f in list(self._listeners):
            f(v)

    def on(self, f):
        self._listeners.append(f)
        return f

    def map(self, f):
        out = Observable(f(self._value))
        self.on(lambda v: setattr(out, "value", f(v)))
        return out


class Processor:
    """`processor(initialize, finalize)` (processors.jl:40-41) in the protocol `solve_unsteady` drives."""

    def __init__(self, initialize, finalize=None, nupdate=1):
        self._initialize, self._finalize = initialize, finalize or (lambda initialized, state: initialized)
        self._obs = None
        # the processor acts only on states whose step count is a multiple of `nupdate` (the library's own factories say so; a user-made processor sees every
        # step): `solve_unsteady` runs the steps in between as one native call (chained steps) instead of one call per step
        self.nupdate = max(1, int(nupdate))

    def initialize(self, getter):
        self._obs = Observable(getter())
        return self._initialize(self._obs)

    def on_step(self, state):
        self._obs.value = state

    def finalize(self, initialized, getter):
        return self._finalize(initialized, getter())


def processor(initialize, finalize=None, nupdate=1):
    return Processor(initialize, finalize, nupdate)


def timelogger(*, showiter=False, showt=True, showdt=True, showmax=True, showspeed=True, nupdate=1, log=print):
    """Create processor that logs time step information (processors.jl:43-76)."""

    def initialize(state):
        told = [state.value["t"]]
        nold = [state.value["n"]]
        oldtime = [time.time()]

        def step(s):
            Δt = (s["t"] - told[0]) / max(1, s["n"] - nold[0])  # the last step's size (the state may arrive every `nupdate` steps only)
            told[0], nold[0] = s["t"], s["n"]
            if s["n"] % nupdate != 0:
                return
            newtime = time.time()
            itertime = (newtime - oldtime[0]) / nupdate
            oldtime[0] = newtime
            msg = []
            if showiter:
                msg.append(f"Iteration {s['n']}")
            if showt:
                msg.append(f"t = {s['t']:g}")
            if showdt:
                msg.append(f"Δt = {Δt:.2g}")
            if showmax:
                msg.append(f"umax = {float(s['u'].abs().max()):.2g}")  # blocking, like maximum(abs, u)
            if showspeed:
                msg.append(f"itertime = {itertime:.2g}")
            log("\t".join(msg))

        state.on(step)
        return None

    return processor(initialize, nupdate=nupdate)


def fieldsaver(*, setup, nupdate=1):
    """Create processor that stores the solution and time every `nupdate` time step (processors.jl:286-300), on the host."""

    def initialize(state):
        states = []

        def step(s):
            if s["n"] % nupdate != 0:
                return
            states.append(dict(u=to_numpy(s["u"]), temp=None if s["temp"] is None else to_numpy(s["temp"]), t=s["t"], n=s["n"]))

        state.on(step)
        return states

    return processor(initialize, nupdate=nupdate)


def observefield(state, *, setup, fieldname, logtol=np.finfo(np.float64).eps, psolver=None):
    """Observe field `fieldname` at pressure points (processors.jl:78-197): an Observable of a host array over Ip, recomputed on the
    device whenever `state` changes.  fieldname: 0/1/2 (velocity component), "velocity", "velocitynorm", "vorticity", "pressure",
    "Dfield", "Qfield", "eig2field", "temperature", "B1".."B11" (tensor-valued), "V1".."V5"."""
    if not isinstance(state, Observable):
        state = Observable(state)
    g = setup.grid
    D = g.dimension
    sl = tuple(slice(lo, hi) for lo, hi in g.Ip)
    up = vectorfield(setup)
    if fieldname == "vorticity":
        ω = scalarfield(setup) if D == 2 else vectorfield(setup)
        ωp = scalarfield(setup) if D == 2 else vectorfield(setup)
    if fieldname in ("pressure", "Dfield"):
        psolver = psolver or default_psolver(setup)
        G, d = vectorfield(setup), scalarfield(setup)
    Q = scalarfield(setup)

    def logclip(f, sign=1.0):
        out = f.clone()
        out[sl] = torch.log(torch.clamp(sign * f[sl], min=logtol))
        return out

    def observe(s):
        u, temp, t = s["u"], s["temp"], s["t"]
        if fieldname in (0, 1, 2):
            f = interpolate_u_p_(up, u, setup)[..., fieldname]
        elif fieldname == "velocity":
            f = interpolate_u_p_(up, u, setup)
        elif fieldname == "velocitynorm":
            f = torch.sqrt((interpolate_u_p_(up, u, setup) ** 2).sum(-1))
        elif fieldname == "vorticity":
            apply_bc_u_(u, t, setup)
            f = interpolate_ω_p_(ωp, vorticity_(ω, u, setup), setup)
        elif fieldname == "pressure":
            f = pressure(u, temp, t, setup, psolver)
        elif fieldname == "Dfield":
            f = logclip(Dfield_(d, G, pressure(u, temp, t, setup, psolver), setup))
        elif fieldname == "Qfield":
            f = logclip(Qfield_(Q, u, setup))
        elif fieldname == "eig2field":
            f = logclip(eig2field_(Q, u, setup), -1.0)
        elif fieldname == "temperature":
            f = temp
        elif isinstance(fieldname, str) and fieldname[:1] in "BV" and fieldname[1:].isdigit():  # "B1".."B11" / "V1".."V5" (processors.jl:121-127)
            from .operators import tensorbasis, tensorbasis_matrices

            B, V = tensorbasis(u, setup)
            idx = int(fieldname[1:]) - 1
            f = V[..., idx] if fieldname[0] == "V" else tensorbasis_matrices(B, setup)[..., idx, :, :]
            return f[sl].cpu().numpy()
        else:
            raise ValueError(f"Unknown fieldname {fieldname!r}")
        return to_numpy(f[sl])

    return state.map(observe)


def spectral_stuff(setup, *, npoint=100, a=(1 + math.sqrt(5)) / 2):
    """Wavenumber shells of the energy spectrum (utils.jl:49-108): `inds[i]` = 0-based column-major positions in the K = Np .÷ 2 array."""
    g = setup.grid
    D = g.dimension
    K = tuple(n // 2 for n in g.Np)
    ks = np.meshgrid(*[np.arange(k, dtype=np.float64) for k in K], indexing="ij")
    k = np.sqrt(sum(x**2 for x in ks)).reshape(-1, order="F")
    kmax = min(K) - 1
    isort = np.argsort(k, kind="stable")
    ksort = k[isort]
    κ = np.unique(np.rint(np.exp(np.linspace(0.0, math.log(kmax), npoint))).astype(np.int64))
    inds = []
    for ki in κ:
        lo, hi = (ki / a, ki * a) if D == 2 else (ki - 0.01, ki + 1 - 0.01)  # dyadic in 2-D, linear in 3-D
        inds.append(isort[np.searchsorted(ksort, lo, side="left") : np.searchsorted(ksort, hi, side="left")])
    return dict(inds=inds, κ=κ, K=K)


class _Spectrum:
    def __init__(self, setup, inds, weights=None):
        self.setup, self.nbin = setup, len(inds)
        off = np.zeros(self.nbin + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(i) for i in inds])
        flat = np.ascontiguousarray(np.concatenate(inds) if off[-1] else np.zeros(1), dtype=np.int64)
        self._handle = C.c_void_p()
        if weights is None:
            _lib.call("ins_spectrum_create", setup.handle, self.nbin, off.ctypes.data_as(C.POINTER(C.c_int64)),
                      flat.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(self._handle))
        else:
            w = np.ascontiguousarray(np.concatenate(weights), dtype=np.float64)
            _lib.call("ins_spectrum_create_weighted", setup.handle, self.nbin, off.ctypes.data_as(C.POINTER(C.c_int64)),
                      flat.ctypes.data_as(C.POINTER(C.c_int64)), w.ctypes.data_as(_lib.c_double_p), C.byref(self._handle))
        _lib.sync_fft_plan_caches()
        self.ehat = torch.zeros(self.nbin, dtype=torch.float64, device=setup.device)

    def __call__(self, u):
        _lib.call("ins_spectrum_f64", self._handle, self.setup.ptr(u, True), C.c_void_p(self.ehat.data_ptr()), self.setup.stream)
        return self.ehat

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            try:
                _lib.load().ins_spectrum_destroy(h)
            except Exception:
                pass


def observespectrum(state, *, setup, npoint=100, a=(1 + math.sqrt(5)) / 2):
    """Observe energy spectrum of `state` (processors.jl:303-332): returns dict(ehat=Observable of a host vector, κ)."""
    if not isinstance(state, Observable):
        state = Observable(state)
    st = spectral_stuff(setup, npoint=npoint, a=a)
    spec = _Spectrum(setup, st["inds"])
    return dict(ehat=state.map(lambda s: spec(s["u"]).cpu().numpy().copy()), κ=st["κ"])


def get_scale_numbers(u, setup):
    """Dimensional scale numbers (operators.jl:1558-1617): uavg, ϵ, η, λ, Reλ, L, τ, Re_int.  Follows the reference's formulas as written —
    including its `uavg`, which sums ALL components' u² under each component's volume weights (D ⟨u_i u_i⟩ on a uniform grid)."""
    from .boundary_conditions import PeriodicBC
    from .operators import dissipation_from_strain

    if not all(isinstance(bc, PeriodicBC) for side in setup.boundary_conditions for bc in side):
        raise ValueError("Scale numbers: the integral length scale needs a uniform periodic grid")  # assert_uniform_periodic, utils.jl:1-13
    g = setup.grid
    D = g.dimension
    visc = 1.0 / setup.Re
    dev = setup.device

    def vec(v, b):
        shape = [1] * D
        shape[b] = len(v)
        return torch.as_tensor(np.asarray(v, dtype=np.float64), device=dev).reshape(shape)

    sl_u = tuple(slice(lo, hi) for lo, hi in g.Iu[0])  # the reference indexes every component with Iu[1]
    sl_p = tuple(slice(lo, hi) for lo, hi in g.Ip)
    uavg2 = 0.0
    for a in range(D):
        Ωu = 1.0
        for b in range(D):
            Ωu = Ωu * vec(g.Δu[b] if a == b else g.Δ[b], b)
        Ωu = Ωu.expand(g.N)
        uavg2 += float(((u[sl_u] ** 2) * Ωu[sl_u].unsqueeze(-1)).sum() / Ωu[sl_u].sum())
    uavg = math.sqrt(uavg2)
    Ω = 1.0
    for b in range(D):
        Ω = Ω * vec(g.Δ[b], b)
    Ω = Ω.expand(g.N)
    ϵf = dissipation_from_strain(u, setup)
    ϵ = float((Ω[sl_p] * ϵf[sl_p]).sum() / Ω[sl_p].sum())
    η = (visc**3 / ϵ) ** 0.25
    λ = math.sqrt(5 * visc / ϵ) * uavg
    Reλ = λ * uavg / math.sqrt(3.0) / visc
    # L = 3π / (2 uavg²) Σ_{k ≠ 0} E(k) / |k| over the retained non-negative wavenumbers (assert_uniform_periodic in the reference)
    K = tuple(n // 2 for n in g.Np)
    ks = np.meshgrid(*[np.arange(k, dtype=np.float64) for k in K], indexing="ij")
    kk = np.sqrt(sum(x**2 for x in ks)).reshape(-1, order="F")
    idx = np.arange(1, kk.size, dtype=np.int64)  # without k = (0, ..., 0)
    spec = _Spectrum(setup, [idx], [1.0 / kk[idx]])
    L = 3 * math.pi / 2 / uavg**2 * float(spec(u)[0])
    # (string keys: Python folds the identifier ϵ to ε, the reference's field name is the lunate form)
    return {"uavg": uavg, "ϵ": ϵ, "η": η, "λ": λ, "Reλ": Reλ, "L": L, "τ": L / uavg, "Re_int": L * uavg / visc}


# ------------------------------------------------------------------------------------ VTK output (processors.jl:199-285)
def _vtk_array(name, a, ncomp=1):
    raw = np.ascontiguousarray(a, dtype=np.float64).tobytes()
    payload = base64.b64encode(np.uint64(len(raw)).tobytes() + raw).decode("ascii")
    return f'<DataArray type="Float64" Name="{name}" NumberOfComponents="{ncomp}" format="binary">{payload}</DataArray>\n'


def save_vtk(state, *, setup, filename="output/solution", fieldnames=("velocity",), psolver=None):
    """Save fields to a VTK rectilinear-grid file `filename.vtr` (processors.jl:199-249): pressure-point coordinates, one point-data
    array per field (2-D vectors get a zero z-component, as ParaView prefers), and the `TimeValue` field datum."""
    if not isinstance(state, Observable):
        state = Observable(state)
    g = setup.grid
    D = g.dimension
    path = os.path.dirname(filename)
    if path:
        os.makedirs(path, exist_ok=True)
    xs = [np.asarray(g.xp[a][g.Ip[a][0] : g.Ip[a][1]], dtype=np.float64) for a in range(D)] + [np.zeros(1)] * (3 - D)
    ext = " ".join(f"0 {len(x) - 1}" for x in xs)
    out = ['<?xml version="1.0"?>\n<VTKFile type="RectilinearGrid" version="1.0" byte_order="LittleEndian" header_type="UInt64">\n',
           f'<RectilinearGrid WholeExtent="{ext}">\n<FieldData>\n', _vtk_array("TimeValue", np.array([state.value["t"]])), "</FieldData>\n",
           f'<Piece Extent="{ext}">\n<PointData>\n']
    for name in fieldnames:
        f = observefield(state, setup=setup, fieldname=name, psolver=psolver).value
        if f.ndim == D + 1:
            comps = [f[..., c] for c in range(D)] + [np.zeros(f.shape[:-1])] * (3 - D)
            data = np.stack([c.reshape(-1, order="F") for c in comps], axis=1)  # point-major, x fastest
            out.append(_vtk_array(str(name), data, 3))
        else:
            out.append(_vtk_array(str(name), f.reshape(-1, order="F")))
    out.append("</PointData>\n<Coordinates>\n")
    for a, x in zip("xyz", xs):
        out.append(_vtk_array(a, x))
    out.append("</Coordinates>\n</Piece>\n</RectilinearGrid>\n</VTKFile>\n")
    fn = filename + ".vtr"
    with open(fn, "w") as fh:
        fh.write("".join(out))
    return fn


def vtk_writer(*, setup, nupdate=1, dir="output", filename="solution", **kwargs):
    """Create processor that writes the solution every `nupdate` time steps to a VTK file and a ParaView collection
    `dir/filename.pvd` (processors.jl:251-285)."""

    def initialize(state):
        os.makedirs(dir, exist_ok=True)
        entries = []

        def step(s):
            if s["n"] % nupdate != 0:
                return
            tformat = str(s["t"]).replace(".", "p")
            fn = save_vtk(s, setup=setup, filename=os.path.join(dir, f"{filename}_t={tformat}"), **kwargs)
            entries.append((s["t"], os.path.basename(fn)))

        state.on(step)
        step(state.value)  # initial step
        return entries

    def finalize(entries, state):
        fn = os.path.join(dir, filename + ".pvd")
        with open(fn, "w") as fh:
            fh.write('<?xml version="1.0"?>\n<VTKFile type="Collection" version="1.0" byte_order="LittleEndian">\n<Collection>\n')
            for t, f in entries:
                fh.write(f'<DataSet timestep="{t}" part="0" file="{f}"/>\n')
            fh.write("</Collection>\n</VTKFile>\n")
        return fn

    return processor(initialize, finalize, nupdate=nupdate)
