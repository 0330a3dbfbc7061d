"""Initial conditions (initializers.jl): evaluated on the host once, projected on the device."""
import numpy as np
import torch

from .operators import apply_bc_temp_, apply_bc_u_
from .pressure import default_psolver, project
from .setup import from_numpy, vectorfield


def velocityfield(setup, ufunc, t=0.0, *, psolver=None, doproject=True):
    """Create divergence free velocity field `u` with boundary conditions at time `t`
    (initializers.jl:13-46).  `ufunc(alpha, x, y[, z])` is called once per component with broadcastable
    coordinate arrays (alpha is 0-based)."""
    g = setup.grid
    D = g.dimension
    host = np.zeros(g.N + (D,), dtype=np.float64, order="F")
    for al in range(D):
        xs = []
        for be in range(D):
            lo, hi = g.Iu[al][be]
            shape = [1] * D
            shape[be] = hi - lo
            xs.append(g.xu[al][be][lo:hi].reshape(shape))
        sl = tuple(slice(lo, hi) for lo, hi in g.Iu[al])
        host[sl + (al,)] = np.broadcast_to(ufunc(al, *xs), tuple(hi - lo for lo, hi in g.Iu[al]))
    u = from_numpy(setup, host)
    apply_bc_u_(u, t, setup)
    if doproject:
        psolver = psolver or default_psolver(setup)
        u = project(u, setup, psolver)
        apply_bc_u_(u, t, setup)
    return u


def temperaturefield(setup, tempfunc, t=0.0):
    """Create temperature field from function with boundary conditions at time `t` (initializers.jl:48-57).
    `tempfunc(x, y[, z])` is called once with broadcastable coordinate arrays of the pressure points."""
    g = setup.grid
    D = g.dimension
    host = np.zeros(g.N, dtype=np.float64, order="F")
    xs = []
    for be in range(D):
        lo, hi = g.Ip[be]
        shape = [1] * D
        shape[be] = hi - lo
        xs.append(np.asarray(g.xp[be][lo:hi]).reshape(shape))
    sl = tuple(slice(lo, hi) for lo, hi in g.Ip)
    host[sl] = np.broadcast_to(tempfunc(*xs), tuple(hi - lo for lo, hi in g.Ip))
    temp = from_numpy(setup, host)
    return apply_bc_temp_(temp, t, setup)


def random_field(setup, t=0.0, *, A=1.0, kp=10, psolver=None, seed=0):
    """Create random field, as in Orlandi 2000 (initializers.jl:189-219, spectrum of :82-181).  The
    reference draws from Julia's Xoshiro stream; this uses torch's Philox generator with `seed`, so fields
    are statistically — not bitwise — equivalent (SURVEY.md §8c vi)."""
    g = setup.grid
    D = g.dimension
    dev = setup.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    tau = 2 * np.pi
    K = tuple((n - 2) // 2 for n in g.N)

    def axis_vec(n, a):
        shape = [1] * D
        shape[a] = n
        return torch.arange(n, dtype=torch.float64, device=dev).reshape(shape)

    def rand(shape):
        return torch.rand(shape, dtype=torch.float64, device=dev, generator=gen)

    k = sum(axis_vec(K[a], a) ** 2 for a in range(D)).sqrt()
    Amag = (8 * tau / 3) / kp**5
    amp = (Amag * k**4 * torch.exp(-tau * (k / kp) ** 2)).sqrt().to(torch.complex128) * float(np.prod(g.N))
    xi = [rand(K) for _ in range(D)]
    for a in range(D):
        amp = torch.cat([amp, torch.flip(amp, dims=[a])], dim=a)
        xi = [torch.cat([xb, torch.flip((-1 if a == b else 1) * xb, dims=[a])], dim=a) for b, xb in enumerate(xi)]
    amp = torch.exp(1j * tau * sum(xi)) * amp
    KK = tuple(2 * kk for kk in K)
    kvec = [axis_vec(KK[a], a).expand(KK) for a in range(D)]
    knorm2 = sum(kv**2 for kv in kvec)
    th = rand(KK)
    if D == 2:
        e = [torch.cos(tau * th), torch.sin(tau * th)]
    else:
        ph = rand(KK)
        e = [torch.sin(np.pi * th) * torch.cos(tau * ph), torch.sin(np.pi * th) * torch.sin(tau * ph), torch.cos(np.pi * th)]
    ke = sum(e[a] * kvec[a] for a in range(D))
    safe = torch.where(knorm2 == 0, torch.ones_like(knorm2), knorm2)
    e = [torch.where(knorm2 == 0, e[a], e[a] - kvec[a] * ke / safe) for a in range(D)]
    enorm = sum(ea**2 for ea in e).sqrt()
    uhat = torch.stack([amp * (ea / enorm) for ea in e], dim=-1)
    uin = A * torch.fft.ifftn(uhat, dim=tuple(range(D))).real
    u = vectorfield(setup)
    inner = tuple(slice(1, n - 1) for n in g.N)
    u[inner] = uin
    apply_bc_u_(u, t, setup)  # periodic ghost fill == NNlib.pad_circular (initializers.jl:209)
    psolver = psolver or default_psolver(setup)
    u = project(u, setup, psolver)
    apply_bc_u_(u, t, setup)
    return u
