"""CPU stand-in for the rank-local slab kernels, built on the oracle — TEST INFRASTRUCTURE ONLY.
Injected into `SlabStepper` by tests/test_dist_gloo.py so the decomposition bookkeeping and the
torch.distributed exchanges can be rehearsed under gloo without a GPU."""
import numpy as np
import torch

from oracle import ins_oracle as o


class OracleSlabKernels:
    def __init__(self, layout, L=(1.0, 1.0, 1.0), Re=1000.0, own=False):
        self.own = own  # mimic the power-of-two path: packed passes, no pack / unpack calls
        self.layout = lay = layout
        nx, ny, nz = lay.n
        self.h = [L[a] / lay.n[a] for a in range(3)]
        x = np.linspace(0.0, L[0], nx + 1)
        y = np.linspace(0.0, L[1], ny + 1)
        z = np.linspace(0.0, L[2], nz + 1)[lay.z0 : lay.z0 + lay.nzl + 1]
        self.setup = o.make_setup((x, y, z), Re=Re)  # periodic metrics == uniform halo metrics
        self.N = self.setup.grid.N
        om = float(np.prod(self.h))
        sym = lambda a, k: 4 * om * np.sin(np.pi * (k / lay.n[a])) ** 2 / self.h[a] ** 2
        self.ax = sym(0, np.arange(lay.kxn))
        self.ay = sym(1, lay.rank * lay.nyl + np.arange(lay.nyl))
        self.az = sym(2, np.arange(nz))
        self.real_elems = nx * ny * lay.nzl
        self.complex_elems = lay.kxn * ny * lay.nzl
        self.cell_volume = om

    # allocation
    def vector(self):
        shape = self.N + (3,)
        return torch.zeros(tuple(reversed(shape)), dtype=torch.float64).permute(3, 2, 1, 0)

    def real(self):
        return torch.zeros(self.real_elems, dtype=torch.float64)

    def cplx(self):
        return torch.zeros(2 * self.complex_elems, dtype=torch.float64)

    def from_global(self, uglob):
        lay = self.layout
        nz = lay.n[2]
        ks = [(lay.z0 + k - 1) % nz + 1 for k in range(lay.nzl + 2)]
        f = self.vector()
        f.numpy()[...] = np.asarray(uglob)[:, :, ks, :]
        return f

    @staticmethod
    def plane(u, c, k):
        return u.permute(3, 2, 1, 0)[c, k]

    def p_plane(self, pI, k):
        n0, n1 = self.layout.n[0], self.layout.n[1]
        return pI[k * n0 * n1 : (k + 1) * n0 * n1]

    # kernels
    @staticmethod
    def _fill_xy(a, planes=slice(None)):
        N0, N1 = a.shape[0], a.shape[1]
        a[0, :, planes] = a[N0 - 2, :, planes]
        a[N0 - 1, :, planes] = a[1, :, planes]
        a[:, 0, planes] = a[:, N1 - 2, planes]
        a[:, N1 - 1, planes] = a[:, 1, planes]

    def fill_xy_ghosts(self, u):
        self._fill_xy(u.numpy())

    def stage_momentum(self, u_in, k_out, ustart, ustar, coefs, ks, coef_self):
        F = o.momentum(np.asfortranarray(u_in.numpy()), None, 0.0, self.setup)
        inner = (slice(1, -1),) * 3
        base = (u_in if ustart is None else ustart).numpy()[inner].copy()
        for cf, k in zip(coefs, ks):
            base = base + cf * k.numpy()[inner]
        base = base + coef_self * F[inner]
        if k_out is not None:
            k_out.numpy()[...] = F
        ustar.numpy()[inner] = base

    def divergence(self, u, pI):
        a = np.asfortranarray(u.numpy().copy())
        self._fill_xy(a)
        d = o.scalewithvolume(o.divergence(a, self.setup), self.setup)
        pI.numpy()[...] = d[1:-1, 1:-1, 1:-1].reshape(-1, order="F")

    def _c(self, buf):
        return buf.numpy().view(np.complex128)

    def fft_forward_xy(self, pI, work, sendbuf):
        lay = self.layout
        nx, ny = lay.n[0], lay.n[1]
        a = pI.numpy().reshape(lay.nzl, ny, nx)  # C-order view of the column-major (nx, ny, nzl) block
        w = np.fft.rfftn(a, axes=(1, 2))  # [kzl][ky][kx]
        packed = w.reshape(lay.nzl, lay.world, lay.nyl, lay.kxn).transpose(1, 0, 2, 3)  # [q][kzl][kyl][kx]
        self._c(sendbuf)[...] = np.ascontiguousarray(packed).reshape(-1)

    def fft_solve_z(self, buf):
        lay = self.layout
        c = self._c(buf).reshape(lay.n[2], lay.nyl, lay.kxn)
        f = np.fft.fft(c, axis=0)
        den = self.az[:, None, None] + self.ay[None, :, None] + self.ax[None, None, :]
        with np.errstate(divide="ignore", invalid="ignore"):
            f = -f / den
        if lay.rank == 0:
            f[0, 0, 0] = 0.0
        c[...] = np.fft.ifft(f, axis=0)

    def fft_inverse_xy(self, recvbuf, work, pI):
        lay = self.layout
        nx, ny = lay.n[0], lay.n[1]
        blocks = self._c(recvbuf).reshape(lay.world, lay.nzl, lay.nyl, lay.kxn)  # [src q][kzl][kyl][kx]
        w = blocks.transpose(1, 0, 2, 3).reshape(lay.nzl, ny, lay.kxn)
        pI.numpy()[...] = np.fft.irfftn(w, s=(ny, nx), axes=(1, 2)).reshape(-1)

    # kx-chunked pieces (same contracts as HipSlabKernels)
    def can_chunk(self):
        return True

    def fft_xy_forward(self, pI, work):
        lay = self.layout
        a = pI.numpy().reshape(lay.nzl, lay.n[1], lay.n[0])
        self._c(work)[...] = np.fft.rfftn(a, axes=(1, 2)).reshape(-1)

    def pack_chunk(self, work, sendbuf, kx0, kxc):
        lay = self.layout
        w = self._c(work).reshape(lay.nzl, lay.world, lay.nyl, lay.kxn)[:, :, :, kx0 : kx0 + kxc]
        self._c(sendbuf)[...] = np.ascontiguousarray(w.transpose(1, 0, 2, 3)).reshape(-1)

    def solve_z_chunk(self, buf, kx0, kxc):
        lay = self.layout
        c = self._c(buf).reshape(lay.n[2], lay.nyl, kxc)
        f = np.fft.fft(c, axis=0)
        den = self.az[:, None, None] + self.ay[None, :, None] + self.ax[None, None, kx0 : kx0 + kxc]
        with np.errstate(divide="ignore", invalid="ignore"):
            f = -f / den
        if lay.rank == 0 and kx0 == 0:
            f[0, 0, 0] = 0.0
        c[...] = np.fft.ifft(f, axis=0)

    def unpack_chunk(self, recvbuf, work, kx0, kxc):
        lay = self.layout
        blocks = self._c(recvbuf).reshape(lay.world, lay.nzl, lay.nyl, kxc)
        w = self._c(work).reshape(lay.nzl, lay.world, lay.nyl, lay.kxn)
        w[:, :, :, kx0 : kx0 + kxc] = blocks.transpose(1, 0, 2, 3)

    def fft_xy_inverse(self, work, pI):
        lay = self.layout
        w = self._c(work).reshape(lay.nzl, lay.n[1], lay.kxn)
        pI.numpy()[...] = np.fft.irfftn(w, s=(lay.n[1], lay.n[0]), axes=(1, 2)).reshape(-1)

    def supports_inkernel(self):
        return self.own and self.layout.nzl >= 2

    def pext(self):
        n0, n1 = self.layout.n[0], self.layout.n[1]
        return torch.zeros(self.real_elems + 3 * n0 * n1, dtype=torch.float64)

    def stage_momentum_corr(self, ustar_prev, p_ext, k_out, ustart, ustar, coefs, ks, coef_self):
        """u = u* - grad p on every padded plane (x, y periodic; z from the exchanged ghost planes), then the plain stage."""
        lay = self.layout
        n0, n1, nzl = lay.n[0], lay.n[1], lay.nzl
        us = ustar_prev.numpy()[1:-1, 1:-1, :, :]
        pp = p_ext.numpy().reshape((n0, n1, nzl + 3), order="F")
        pc = pp[:, :, : nzl + 2]
        corr = np.zeros(self.N + (3,), order="F")
        corr[1:-1, 1:-1, :, 0] = us[..., 0] - (np.roll(pp, -1, axis=0)[:, :, : nzl + 2] - pc) / self.h[0]
        corr[1:-1, 1:-1, :, 1] = us[..., 1] - (np.roll(pp, -1, axis=1)[:, :, : nzl + 2] - pc) / self.h[1]
        corr[1:-1, 1:-1, :, 2] = us[..., 2] - (pp[:, :, 1 : nzl + 3] - pc) / self.h[2]
        self._fill_xy(corr)
        t = self.vector()
        t.numpy()[...] = corr
        self.stage_momentum(t, k_out, ustart, ustar, coefs, ks, coef_self)

    def is_own(self):
        return self.own

    def fft_forward_packed(self, u, work, sendbuf, cw):
        lay = self.layout
        pI = self.real()
        self.divergence(u, pI)
        self.fft_xy_forward(pI, work)
        per_kx = 2 * lay.world * lay.nzl * lay.nyl
        for k0 in range(0, lay.kxn, cw):
            kc = min(cw, lay.kxn - k0)
            self.pack_chunk(work, sendbuf[per_kx * k0 : per_kx * (k0 + kc)], k0, kc)

    def fft_inverse_packed(self, recvbuf, work, pI, cw):
        lay = self.layout
        per_kx = 2 * lay.world * lay.nzl * lay.nyl
        for k0 in range(0, lay.kxn, cw):
            kc = min(cw, lay.kxn - k0)
            self.unpack_chunk(recvbuf[per_kx * k0 : per_kx * (k0 + kc)], work, k0, kc)
        self.fft_xy_inverse(work, pI)

    # ---- transpose-free z solve: independent numpy restatement of csrc/ins_ztri.hip's contract ------------------------
    # (dense per-line Thomas with explicit pivots, explicit spikes from two more Thomas solves, dense 2P x 2P interface
    #  systems, np.fft for the singular line — none of the closed forms the HIP kernels use)
    def supports_ztri(self):
        return self.layout.nzl >= 2

    def _ztri_setup(self):
        lay = self.layout
        ny = lay.n[1]
        om = self.cell_volume
        self.cz = om / self.h[2] ** 2
        ayf = 4 * om * np.sin(np.pi * (np.arange(ny) / ny)) ** 2 / self.h[1] ** 2
        self.dline = ayf[:, None] + self.ax[None, :] + 2 * self.cz  # d = âx + ây + 2c, [ky][kx]
        self.lines = ny * lay.kxn

    def ztri_edge(self, ranks=1):
        if not hasattr(self, "lines"):
            self._ztri_setup()
        return torch.zeros(2 * (2 * self.lines + self.layout.nzl) * ranks, dtype=torch.float64)

    def _thomas(self, g):
        """A_r^{-1} g for every line: g [m][ky][kx] complex, A_r = tridiag(-c, d, -c)."""
        m, c, d = g.shape[0], self.cz, self.dline
        cp = np.zeros((m,) + d.shape)
        gp = np.zeros_like(g)
        den = d.copy()
        gp[0] = g[0] / den
        for k in range(1, m):
            cp[k - 1] = -c / den
            den = d + c * cp[k - 1]
            gp[k] = (g[k] + c * gp[k - 1]) / den
        y = np.zeros_like(g)
        y[m - 1] = gp[m - 1]
        for k in range(m - 2, -1, -1):
            y[k] = gp[k] - cp[k] * y[k + 1]
        return y

    def ztri_forward(self, src, from_u, work, edge):
        lay = self.layout
        if not hasattr(self, "lines"):
            self._ztri_setup()
        pI = src
        if from_u:
            pI = self.real()
            self.divergence(src, pI)
        a = pI.numpy().reshape(lay.nzl, lay.n[1], lay.n[0])
        g = -np.fft.rfftn(a, axes=(1, 2))  # [kzl][ky][kx]; numpy's irfftn carries the 1/(nx ny) the HIP passes fold into g
        e = self._c(edge)
        e[2 * self.lines :] = g[:, 0, 0]
        g[:, 0, 0] = 0.0
        y = self._thomas(g)
        self._c(work)[...] = y.reshape(-1)  # the stand-in keeps y = A_r^{-1} g (the HIP kernel keeps the half-eliminated field)
        e[: self.lines] = y[0].reshape(-1)
        e[self.lines : 2 * self.lines] = y[-1].reshape(-1)

    def ztri_finish(self, work, edges_all, pI):
        lay = self.layout
        P, m, c = lay.world, lay.nzl, self.cz
        ny, kxn = lay.n[1], lay.kxn
        ea = self._c(edges_all).reshape(P, 2 * self.lines + m)
        yF = ea[:, : self.lines].reshape(P, ny, kxn)
        yL = ea[:, self.lines : 2 * self.lines].reshape(P, ny, kxn)
        e0 = np.zeros((m, ny, kxn), dtype=complex)
        e0[0] = c
        v = self._thomas(e0).real  # c A_r^{-1} e_first
        w = v[::-1]                # c A_r^{-1} e_last (symmetric Toeplitz)
        # interface: unknowns z = [F_0, L_0, F_1, L_1, ...]
        M = np.zeros((ny, kxn, 2 * P, 2 * P))
        rhs = np.zeros((ny, kxn, 2 * P), dtype=complex)
        for r in range(P):
            rp, rn = (r - 1) % P, (r + 1) % P
            M[..., 2 * r, 2 * r] += 1.0
            M[..., 2 * r, 2 * rp + 1] += -v[0]
            M[..., 2 * r, 2 * rn] += -w[0]
            M[..., 2 * r + 1, 2 * r + 1] += 1.0
            M[..., 2 * r + 1, 2 * rp + 1] += -v[m - 1]
            M[..., 2 * r + 1, 2 * rn] += -w[m - 1]
            rhs[..., 2 * r] = yF[r]
            rhs[..., 2 * r + 1] = yL[r]
        M[0, 0] = np.eye(2 * P)  # the singular line is handled below
        z = np.linalg.solve(M, rhs[..., None])[..., 0]
        Lprev = z[..., 2 * ((lay.rank - 1) % P) + 1]
        Fnext = z[..., 2 * ((lay.rank + 1) % P)]
        y = self._c(work).reshape(m, ny, kxn)
        p = y + Lprev[None] * v + Fnext[None] * w
        # singular line: periodic 1-D Poisson over all nz values, zero mean (pressure.jl:336-341)
        g0 = ea[:, 2 * self.lines :].reshape(-1)
        gh = np.fft.fft(g0)
        gh[1:] /= self.az[1:]
        gh[0] = 0.0
        p[:, 0, 0] = np.fft.ifft(gh)[lay.rank * m : (lay.rank + 1) * m]
        pI.numpy()[...] = np.fft.irfftn(p, s=(ny, lay.n[0]), axes=(1, 2)).reshape(-1)

    def applypressure(self, u, pI, p_top):
        lay = self.layout
        nx, ny = lay.n[0], lay.n[1]
        p = np.zeros(self.N, order="F")
        p[1:-1, 1:-1, 1:-1] = pI.numpy().reshape((nx, ny, lay.nzl), order="F")
        p[1:-1, 1:-1, -1] = p_top.numpy().reshape((nx, ny), order="F")
        self._fill_xy(p)
        a = u.numpy()
        o.applypressure_(a, p, self.setup)
        self._fill_xy(a, planes=slice(1, lay.nzl + 1))

    def sync(self):
        pass
