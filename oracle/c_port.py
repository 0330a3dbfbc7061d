"""ctypes driver of oracle/c/ins_oracle_c.c — the multi-threaded CPU restatement (TEST / BASELINE ONLY).
One RK step for periodic 3-D boxes with the reference's unfused pass structure
(step_explicit_runge_kutta.jl:4-59); FFTs through scipy's pocketfft with `workers` threads."""
import ctypes as C
import os

import numpy as np
import scipy.fft

from . import ins_oracle as o

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "c", "libins_oracle_c.so")
dp = C.POINTER(C.c_double)


def _ptr(a):
    return a.ctypes.data_as(dp)


class CPort:
    def __init__(self, setup, workers=None):
        if not os.path.exists(_LIB):
            raise RuntimeError(f"{_LIB} missing: run `make -C oracle/c`")
        self.lib = C.CDLL(_LIB)
        self.s = setup
        g = setup.grid
        assert g.D == 3
        self.N = g.N
        self.workers = workers or os.cpu_count()
        self._keep = []

        def arr3(vs):
            vs = [np.ascontiguousarray(v, dtype=np.float64) for v in vs]
            self._keep.extend(vs)
            return (dp * len(vs))(*[_ptr(v) for v in vs])

        self.dx, self.dxu = arr3(g.dx), arr3(g.dxu)
        self.A1 = arr3([g.A[b][a][0] for b in range(3) for a in range(3)])
        self.A2 = arr3([g.A[b][a][1] for b in range(3) for a in range(3)])
        self.ahat = [np.ascontiguousarray(a) for a in o.spectral_symbols(setup)]
        n = g.Np
        self.pI = np.zeros(n[0] * n[1] * n[2])
        self.nvec = int(np.prod(self.N)) * 3

    def bc_u(self, u):
        self.lib.oc_bc_periodic(_ptr(u), *self.N, 3)

    def momentum(self, F, u):
        self.lib.oc_momentum(_ptr(F), _ptr(u), C.c_double(1.0 / self.s.Re), *self.N, self.dx, self.dxu, self.A1, self.A2)

    def project(self, u, p):
        N, n = self.N, self.s.grid.Np
        self.lib.oc_divergence(_ptr(p), _ptr(u), *N, self.dx)
        self.lib.oc_scalewithvolume(_ptr(p), *N, self.dx)
        self.lib.oc_strip(_ptr(self.pI), _ptr(p), *N)
        a = self.pI.reshape(n[2], n[1], n[0])
        ph = scipy.fft.rfftn(a, workers=self.workers)
        phv = np.ascontiguousarray(ph).view(np.float64)
        self.lib.oc_symbol(_ptr(phv), _ptr(self.ahat[0]), _ptr(self.ahat[1]), _ptr(self.ahat[2]), n[0] // 2 + 1, n[1], n[2])
        back = scipy.fft.irfftn(phv.view(np.complex128).reshape(ph.shape), s=a.shape, workers=self.workers)
        self.pI[...] = back.reshape(-1)
        self.lib.oc_pad(_ptr(p), _ptr(self.pI), *N)
        self.lib.oc_bc_periodic(_ptr(p), *N, 1)
        self.lib.oc_applypressure(_ptr(u), _ptr(p), *N, self.dxu)

    def timestep_(self, method, u, dt, cache):
        """In-place RK step on the flat Fortran-ordered array `u` (shape N+(3,))."""
        A = method.A
        ns = len(method.b)
        ustart, ku, p = cache["ustart"], cache["ku"], cache["p"]
        n = C.c_size_t(self.nvec)
        self.lib.oc_copy(_ptr(ustart), _ptr(u), n)
        for i in range(ns):
            self.bc_u(u)
            self.momentum(ku[i], u)
            self.lib.oc_copy(_ptr(u), _ptr(ustart), n)
            for j in range(i + 1):  # tableau zeros are not skipped (step_explicit_runge_kutta.jl:36-38)
                self.lib.oc_axpy(_ptr(u), C.c_double(dt * A[i, j]), _ptr(ku[j]), n)
            self.bc_u(u)
            self.project(u, p)
        self.bc_u(u)
        return u
