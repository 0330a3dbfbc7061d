"""Multi-GPU z-slab path: one process per GPU, exchanges over torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" for CPU rehearsals).

The reference has no multi-device code; this module distributes exactly the reference's stage loop
(step_explicit_runge_kutta.jl:17-50) over ranks:

    rank r owns interior z-planes [r*nz/P, (r+1)*nz/P) (+ one ghost plane per side); x, y stay whole.
    per stage:  halo(u)  ->  K1+K6 (k_i, u*)  ->  halo(w-plane of u*)  ->  K2 (Ω div u* -> pI)
                -> 2-D FFT(x,y)
                   zsolve = "tridiag" (default for > 1 rank): forward elimination along z + interface data
                      -> all-gather (2 complex numbers per (kx, ky) line) -> back substitution     [csrc/ins_ztri.hip]
                   zsolve = "fft": pack -> all-to-all -> z-FFT / symbol / inverse z-FFT -> all-to-all -> unpack
                -> inverse 2-D FFT -> halo(first p plane) -> K4 (u* -= ∇p, x/y ghost images)

The z direction of the spectral solve is a periodic tridiagonal system per (kx, ky) line (the circulant matrix the z-FFT
diagonalises).  Solving it across ranks by the partition method moves ~1 MB per rank and solve instead of the whole half
spectrum twice (135 MB per rank and transpose at 256^3 per rank), which is what xGMI's point-to-point links cannot hide.

`SlabStepper` holds only index bookkeeping and communication; the rank-local numerics come from a
`kernels` object: `HipSlabKernels` below (libinship, the product) — tests inject a CPU implementation to
rehearse the communication pattern under gloo without a GPU.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .boundary_conditions import HaloBC, PeriodicBC
from .setup import Setup, vectorfield


class SlabLayout:
    """Index bookkeeping of the z-slab decomposition of an (nx, ny, nz) periodic box over `world` ranks."""

    def __init__(self, n, world, rank):
        self.n = tuple(int(v) for v in n)
        self.world, self.rank = int(world), int(rank)
        nx, ny, nz = self.n
        if nz % world or ny % world:
            raise ValueError(f"nz={nz} and ny={ny} must be divisible by the number of ranks ({world})")
        self.nzl = nz // world
        self.nyl = ny // world
        self.z0 = rank * self.nzl
        self.kxn = nx // 2 + 1
        self.prev = (rank - 1) % world
        self.next = (rank + 1) % world

    @property
    def local_shape(self):  # padded local field shape
        return (self.n[0] + 2, self.n[1] + 2, self.nzl + 2)


class SlabComm:
    """Point-to-point plane exchange and the transpose all-to-all over a torch.distributed group.
    With the gloo backend and device tensors the payload is staged through the host (rehearsal mode)."""

    def __init__(self, group=None, group2=None, loopback=False):
        """loopback: with ONE rank, still run every exchange through the process group (sends to self) instead of the local copy — the
        one-GPU rehearsal of the RCCL calls, their matching order when both neighbours are the same peer, and their stream ordering
        (tests/test_gpu_nccl.py)."""
        self.loopback = bool(loopback)
        self.group = group
        self.group2 = group2 if group2 is not None else group  # second communicator: back-transposes run beside forward ones
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"

    def _stage(self, t):
        return self.backend == "gloo" and t.is_cuda

    def exchange(self, sends, recvs):
        """sends: [(tensor, dst)], recvs: [(tensor, src)] — all contiguous, matched pairwise across ranks."""
        if self.world == 1 and not self.loopback:
            for (s, _), (r, _) in zip(sends, recvs):
                r.copy_(s)
            return
        stage = any(self._stage(t) for t, _ in sends)
        if stage:
            hs = [(t.cpu(), d) for t, d in sends]
            hr = [(torch.empty(t.shape, dtype=t.dtype), s) for t, s in recvs]
        else:
            hs, hr = sends, recvs
        ops = [dist.P2POp(dist.isend, t, d, self.group) for t, d in hs] + [dist.P2POp(dist.irecv, t, s, self.group) for t, s in hr]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if stage:
            for (t, _), (h, _) in zip(recvs, hr):
                t.copy_(h)

    def exchange_async(self, sends, recvs):
        """Like exchange(), but returns the outstanding requests (RCCL: the transfer runs beside later kernels of the current
        stream; `wait()` orders the stream after it).  Staged / single-rank modes complete immediately."""
        if (self.world == 1 and not self.loopback) or any(self._stage(t) for t, _ in sends):
            self.exchange(sends, recvs)
            return []
        ops = [dist.P2POp(dist.isend, t, d, self.group) for t, d in sends] + [dist.P2POp(dist.irecv, t, s, self.group) for t, s in recvs]
        return dist.batch_isend_irecv(ops)

    def all_to_all(self, recv, send):
        if self.world == 1 and not self.loopback:
            recv.copy_(send)
            return
        if self._stage(send):
            hs, hr = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_to_all_single(hr, hs, group=self.group)
            recv.copy_(hr)
        else:
            dist.all_to_all_single(recv, send, group=self.group)

    class _Done:
        def wait(self):
            return None

    def all_to_all_async(self, recv, send, which=0):
        """Start a transpose; returns a handle whose wait() orders the current stream after it (no host block on RCCL)."""
        if (self.world == 1 and not self.loopback) or self._stage(send):
            self.all_to_all(recv, send)
            return SlabComm._Done()
        return dist.all_to_all_single(recv, send, group=self.group2 if which else self.group, async_op=True)

    def all_gather(self, out, inp):
        """out[r*len(inp):(r+1)*len(inp)] = rank r's inp.
        On RCCL the gather is done as direct sends to every peer in one group call: xGMI is a full mesh of point-to-point links,
        so each ~1 MB block crosses exactly one link and all links work at once, instead of the P-1 dependent hops of a ring
        (INS_SLAB_GATHER=collective selects all_gather_into_tensor)."""
        if self.world == 1 and not self.loopback:
            out.copy_(inp)
            return
        n = inp.numel()
        if self._stage(inp):
            ho = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(ho, inp.cpu(), group=self.group)
            out.copy_(ho)
        elif (os.environ.get("INS_SLAB_GATHER") or ("p2p" if self.backend == "nccl" else "collective")) == "p2p":
            out[self.rank * n : (self.rank + 1) * n].copy_(inp)
            peers = [(self.rank + d) % self.world for d in range(1, self.world)]
            ops = [dist.P2POp(dist.isend, inp, q, self.group) for q in peers]
            ops += [dist.P2POp(dist.irecv, out[q * n : (q + 1) * n], q, self.group) for q in reversed(peers)]
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
        else:
            dist.all_gather_into_tensor(out, inp, group=self.group)

    def all_gather_async(self, out, inp):
        """all_gather whose transfer runs beside later kernels of the current stream; returns requests to wait() on ([] when it
        completed synchronously: one rank, staged rehearsal or the collective route)."""
        mode = os.environ.get("INS_SLAB_GATHER") or ("p2p" if self.backend == "nccl" else "collective")
        if (self.world == 1 and not self.loopback) or self._stage(inp) or mode != "p2p":
            self.all_gather(out, inp)
            return []
        n = inp.numel()
        out[self.rank * n : (self.rank + 1) * n].copy_(inp)
        peers = [(self.rank + d) % self.world for d in range(1, self.world)]
        ops = [dist.P2POp(dist.isend, inp, q, self.group) for q in peers]
        ops += [dist.P2POp(dist.irecv, out[q * n : (q + 1) * n], q, self.group) for q in reversed(peers)]
        return dist.batch_isend_irecv(ops) if ops else []

    def barrier(self):
        if self.world > 1:
            dist.barrier(self.group)


class AbiSlabComm:
    """The SlabComm interface on the library's OWN RCCL entry points (`ins_comm_*`, csrc/ins_comm.hip; include/ins_hip.h) instead of
    torch.distributed — the route a Julia host takes.  One communicator per process / GPU:

        id = AbiSlabComm.unique_id()            # rank 0; ship the 128 bytes to the other ranks (MPI, a file, torch.distributed ...)
        comm = AbiSlabComm(world, rank, id)     # every rank, its device current

    Exchanges are enqueued on the current stream; the asynchronous variants run on a side stream ordered by events (the RCCL kernels then
    overlap the compute kernels) and return handles whose wait() orders the current stream after them."""

    backend = "rccl-abi"

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        _lib.call("ins_comm_unique_id", buf)
        return bytes(buf)

    def __init__(self, world, rank, id_bytes, device=None, loopback=False, overlap=True):
        self.world, self.rank, self.loopback = int(world), int(rank), bool(loopback)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.call("ins_comm_create", self.world, self.rank, C.create_string_buffer(bytes(id_bytes), 128), C.byref(self._h))
        self.side = torch.cuda.Stream(self.device) if overlap else None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.load().ins_comm_destroy(h)
            except Exception:
                pass

    class _After:
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    def _local(self):
        return self.world == 1 and not self.loopback

    def _stream_ptr(self, stream=None):
        return C.c_void_p((stream or torch.cuda.current_stream(self.device)).cuda_stream)

    def _sendrecv(self, sends, recvs, stream=None):
        ns, nr = len(sends), len(recvs)
        sp = (C.c_void_p * max(ns, 1))(*[t.data_ptr() for t, _ in sends])
        sc = (C.c_int64 * max(ns, 1))(*[t.numel() for t, _ in sends])
        sd = (C.c_int32 * max(ns, 1))(*[d for _, d in sends])
        rp = (C.c_void_p * max(nr, 1))(*[t.data_ptr() for t, _ in recvs])
        rc = (C.c_int64 * max(nr, 1))(*[t.numel() for t, _ in recvs])
        rs = (C.c_int32 * max(nr, 1))(*[q for _, q in recvs])
        _lib.call("ins_comm_sendrecv_f64", self._h, ns, sp, sc, sd, nr, rp, rc, rs, self._stream_ptr(stream))

    def _on_side(self, fn):
        """Run fn(stream) on the side stream behind everything enqueued so far; returns a handle to wait() on."""
        if self.side is None:
            fn(None)
            return []
        cur = torch.cuda.current_stream(self.device)
        self.side.wait_stream(cur)
        fn(self.side)
        ev = torch.cuda.Event()
        ev.record(self.side)
        return [AbiSlabComm._After(ev)]

    def exchange(self, sends, recvs):
        if self._local():
            for (s, _), (r, _) in zip(sends, recvs):
                r.copy_(s)
            return
        self._sendrecv(sends, recvs)

    def exchange_async(self, sends, recvs):
        if self._local():
            self.exchange(sends, recvs)
            return []
        return self._on_side(lambda st: self._sendrecv(sends, recvs, st))

    def all_to_all(self, recv, send, stream=None):
        if self._local():
            recv.copy_(send)
            return
        _lib.call("ins_comm_alltoall_f64", self._h, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), send.numel() // self.world, self._stream_ptr(stream))

    def all_to_all_async(self, recv, send, which=0):
        if self._local():
            self.all_to_all(recv, send)
            return SlabComm._Done()
        h = self._on_side(lambda st: self.all_to_all(recv, send, st))
        return h[0] if h else SlabComm._Done()

    def all_gather(self, out, inp, stream=None):
        if self._local():
            out.copy_(inp)
            return
        direct = 0 if os.environ.get("INS_SLAB_GATHER") == "collective" else 1
        _lib.call("ins_ztri_allgather_f64", self._h, C.c_void_p(inp.data_ptr()), C.c_void_p(out.data_ptr()), inp.numel(), direct, self._stream_ptr(stream))

    def all_gather_async(self, out, inp):
        if self._local():
            self.all_gather(out, inp)
            return []
        return self._on_side(lambda st: self.all_gather(out, inp, st))

    def allreduce_(self, t, op="max"):
        """In-place reduction of a device fp64 tensor across ranks (op: sum / max / min)."""
        if not self._local():
            _lib.call("ins_comm_allreduce_f64", self._h, C.c_void_p(t.data_ptr()), t.numel(), {"sum": 0, "max": 1, "min": 2}[op], self._stream_ptr())
        return t

    def halo_u(self, setup, u, comps=(0, 1, 2), down_only=False):
        """`ins_halo_exchange_f64`: the z ghost planes of a padded local vector field in one call (what SlabStepper.halo_u assembles plane by plane)."""
        mask = sum(1 << c for c in comps)
        _lib.call("ins_halo_exchange_f64", self._h, setup.handle, setup.ptr(u, True), mask, 1 if down_only else 0, self._stream_ptr())

    def barrier(self):
        if not self._local():
            t = torch.zeros(1, dtype=torch.float64, device=self.device)
            self.allreduce_(t, "sum")
            torch.cuda.synchronize(self.device)


class HipSlabKernels:
    """Rank-local numerics of the slab path: HIP kernels + rocFFT through the C ABI (include/ins_hip.h)."""

    def __init__(self, layout, L=(1.0, 1.0, 1.0), Re=1000.0, device=None):
        self.layout = lay = layout
        nx, ny, nz = lay.n
        h = [L[a] / lay.n[a] for a in range(3)]
        x = np.linspace(0.0, L[0], nx + 1)
        y = np.linspace(0.0, L[1], ny + 1)
        z = np.linspace(0.0, L[2], nz + 1)[lay.z0 : lay.z0 + lay.nzl + 1]
        per = (PeriodicBC(), PeriodicBC())
        self.setup = Setup(x=(x, y, z), boundary_conditions=(per, per, (HaloBC(), HaloBC())), Re=Re, device=device)
        self.device = self.setup.device
        np3 = (C.c_int32 * 3)(nx, ny, nz)
        h3 = (C.c_double * 3)(*h)
        self._fft = C.c_void_p()
        _lib.call("ins_slab_fft_create", np3, h3, lay.rank, lay.world, C.byref(self._fft))
        _lib.sync_fft_plan_caches()
        nr, nc = C.c_int64(), C.c_int64()
        _lib.call("ins_slab_fft_sizes", self._fft, C.byref(nr), C.byref(nc))
        self.real_elems, self.complex_elems = nr.value, nc.value
        self.cell_volume = float(h[0] * h[1] * h[2])

    def __del__(self):
        h, self._fft = getattr(self, "_fft", None), None
        if h:
            try:
                _lib.load().ins_slab_fft_destroy(h)
            except Exception:
                pass

    # ---- allocation -------------------------------------------------------------------------
    def vector(self):
        return vectorfield(self.setup)

    def real(self):
        return torch.zeros(self.real_elems, dtype=torch.float64, device=self.device)

    def cplx(self):
        return torch.zeros(2 * self.complex_elems, dtype=torch.float64, device=self.device)

    def from_global(self, uglob):
        """Cut this rank's slab (with z ghost planes, periodic) out of a global padded host field."""
        lay = self.layout
        nz = lay.n[2]
        ks = [(lay.z0 + k - 1) % nz + 1 for k in range(lay.nzl + 2)]  # global padded plane indices
        local = np.asfortranarray(np.asarray(uglob)[:, :, ks, :])
        f = self.vector()
        f.copy_(torch.from_numpy(np.ascontiguousarray(local)).to(self.device))
        return f

    # ---- views used by the exchanges --------------------------------------------------------
    @staticmethod
    def plane(u, c, k):
        """Contiguous (N1*N0) view of component c, padded z-plane k of a local vector field."""
        return u.permute(3, 2, 1, 0)[c, k]

    def p_plane(self, pI, k):
        n0, n1 = self.layout.n[0], self.layout.n[1]
        return pI[k * n0 * n1 : (k + 1) * n0 * n1]

    # ---- kernels ----------------------------------------------------------------------------
    def _p(self, t):
        return C.c_void_p(t.data_ptr())

    def fill_xy_ghosts(self, u):
        s = self.setup
        _lib.call("ins_apply_bc_u_f64", s.handle, s.ptr(u, True), 0, None, s.stream)

    def stage_momentum(self, u_in, k_out, ustart, ustar, coefs, ks, coef_self):
        s = self.setup
        n = len(coefs)
        carr = (C.c_double * max(n, 1))(*coefs)
        karr = (C.c_void_p * max(n, 1))(*[k.data_ptr() for k in ks])
        _lib.call("ins_stage_momentum_f64", s.handle, 1.0 / s.Re, s.ptr(u_in, True), s.ptr(k_out, True) if k_out is not None else None,
                  s.ptr(ustart, True) if ustart is not None else None, s.ptr(ustar, True), n, carr, karr, float(coef_self), s.stream)

    def supports_inkernel(self):
        """Stages >= 2 can apply the previous projection in registers (exactly-uniform slab with >= 2 local planes)."""
        return bool(_lib.load().ins_grid_is_uniform_exact(self.setup.handle)) and self.layout.nzl >= 2 and self.is_own()

    def pext(self):
        """Extended pressure buffer [1 plane below | nzl local planes | 2 planes above], unpadded in x, y."""
        n0, n1 = self.layout.n[0], self.layout.n[1]
        return torch.zeros(self.real_elems + 3 * n0 * n1, dtype=torch.float64, device=self.device)

    def stage_momentum_corr(self, ustar_prev, p_ext, k_out, ustart, ustar, coefs, ks, coef_self, part=0, c0m1=0.0, self_in=0.0, ustart_out=None):
        """part 1: the planes that read no ghost plane (can run beside the halo exchange); part 2: the rest; 0: everything.
        c0m1 != 0: `ks` are earlier uncorrected stage velocities (stage-velocity basis, see SlabStepper)."""
        s = self.setup
        n = len(coefs)
        carr = (C.c_double * max(n, 1))(*coefs)
        karr = (C.c_void_p * max(n, 1))(*[k.data_ptr() for k in ks])
        prof = getattr(self, "prof", None)
        if prof is not None:  # measurement hook (bench_dist.py `roofline`): events on the stream the kernel is launched on
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.call("ins_stage_momentum_corr_part_f64", s.handle, 1.0 / s.Re, s.ptr(ustar_prev, True), self._p(p_ext),
                  s.ptr(k_out, True) if k_out is not None else None, s.ptr(ustart, True) if ustart is not None else None, s.ptr(ustar, True), n, carr,
                  karr, float(coef_self), float(c0m1), float(self_in), s.ptr(ustart_out, True) if ustart_out is not None else None, int(part), s.stream)
        if prof is not None:
            e1.record()
            # algorithmic bytes per cell of this stage (counted once, with its first part): R u*, R p, W out, R ustart, R earlier stage fields, ...
            b = 0 if part == 2 else 24 * (2 + (ustart is not None) + n + (k_out is not None) + (ustart_out is not None)) + 8
            prof.append((e0, e1, b))

    # stage_momentum_corr(part=...) and xfwd_planes exist: SlabStepper overlaps exchanges with them (INS_SLAB_NO_SPLIT=1: A/B switch)
    splits_stage = not bool(os.environ.get("INS_SLAB_NO_SPLIT"))

    def wide_stage_kernel(self):
        """The 64-outputs-per-wavefront stage kernel runs on this slab: the stencil input can then be a term of the combination."""
        return (bool(_lib.load().ins_grid_is_uniform_exact(self.setup.handle)) and self.layout.n[0] >= 66 and self.layout.n[1] >= 8
                and self.layout.nzl >= 4 and not os.environ.get("INS_DISABLE_FLUX64"))

    def xfwd_planes(self, u, work, kz0, nkz):
        s = self.setup
        _lib.call("ins_slab_xfwd_planes", self._fft, s.handle, s.ptr(u, True), self._p(work), int(kz0), int(nkz), s.stream)

    def divergence(self, u, pI):
        s = self.setup
        _lib.call("ins_slab_divergence_f64", s.handle, s.ptr(u, True), self._p(pI), s.stream)

    def fft_forward_xy(self, pI, work, sendbuf):
        _lib.call("ins_slab_fft_forward_xy", self._fft, self._p(pI), self._p(work), self._p(sendbuf), self.setup.stream)

    def fft_solve_z(self, buf):
        _lib.call("ins_slab_fft_solve_z", self._fft, self._p(buf), self.setup.stream)

    def fft_inverse_xy(self, recvbuf, work, pI):
        _lib.call("ins_slab_fft_inverse_xy", self._fft, self._p(recvbuf), self._p(work), self._p(pI), self.setup.stream)

    # kx-chunked pieces
    def can_chunk(self):
        return bool(_lib.load().ins_slab_fft_can_chunk(self._fft))

    def fft_xy_forward(self, pI, work):
        _lib.call("ins_slab_fft_xy_forward_only", self._fft, self._p(pI), self._p(work), self.setup.stream)

    def pack_chunk(self, work, sendbuf, kx0, kxc):
        _lib.call("ins_slab_fft_pack_chunk", self._fft, self._p(work), self._p(sendbuf), kx0, kxc, self.setup.stream)

    def solve_z_chunk(self, buf, kx0, kxc):
        _lib.call("ins_slab_fft_solve_z_chunk", self._fft, self._p(buf), kx0, kxc, self.setup.stream)

    def unpack_chunk(self, recvbuf, work, kx0, kxc):
        _lib.call("ins_slab_fft_unpack_chunk", self._fft, self._p(recvbuf), self._p(work), kx0, kxc, self.setup.stream)

    def fft_xy_inverse(self, work, pI):
        _lib.call("ins_slab_fft_xy_inverse_only", self._fft, self._p(work), self._p(pI), self.setup.stream)

    # power-of-two boxes: passes that write / read the packed exchange buffer directly
    def is_own(self):
        return bool(_lib.load().ins_slab_fft_is_own(self._fft))

    def fft_forward_packed(self, u, work, sendbuf, cw):
        """Ω·div(u) formed inside the x pass (K2 fused), y pass writes the packed, kx-chunked send buffer."""
        s = self.setup
        _lib.call("ins_slab_fft_forward_packed", self._fft, s.handle, s.ptr(u, True), 1, self._p(work), self._p(sendbuf), cw, s.stream)

    def fft_inverse_packed(self, recvbuf, work, pI, cw):
        _lib.call("ins_slab_fft_inverse_packed", self._fft, self._p(recvbuf), self._p(work), self._p(pI), cw, self.setup.stream)

    # transpose-free z solve (csrc/ins_ztri.hip)
    def supports_ztri(self):
        return self.layout.nzl >= 2 and self.layout.world <= 16

    def ztri_edge(self, ranks=1):
        n = C.c_int64()
        _lib.call("ins_slab_ztri_edge_elems", self._fft, C.byref(n))
        return torch.zeros(n.value * ranks, dtype=torch.float64, device=self.device)

    def ztri_forward(self, src, from_u, work, edge):
        """(x, y) transforms of the local planes (from_u = 1: Ω·div(u) formed inside the x pass; 2: the x pass was done by
        xfwd_planes), forward elimination along z in place on `work`, this rank's interface data -> edge."""
        s = self.setup
        ptr = s.ptr(src, True) if from_u else self._p(src)
        _lib.call("ins_slab_ztri_forward", self._fft, s.handle, ptr, int(from_u), self._p(work), self._p(edge), s.stream)

    def ztri_finish(self, work, edges_all, pI):
        _lib.call("ins_slab_ztri_finish", self._fft, self._p(work), self._p(edges_all), self._p(pI), self.setup.stream)

    # the same solve with the lines cut into ranges (the gather of one range travels while the next range is swept)
    ztri_chunked = True

    def ztri_chunk_edge(self, c, nchunks, ranks=1):
        n = C.c_int64()
        _lib.call("ins_slab_ztri_chunk", self._fft, int(c), int(nchunks), None, None, C.byref(n))
        return torch.zeros(n.value * ranks, dtype=torch.float64, device=self.device)

    def ztri_transform(self, src, from_u, work):
        s = self.setup
        ptr = s.ptr(src, True) if from_u else self._p(src)
        _lib.call("ins_slab_ztri_transform", self._fft, s.handle, ptr, int(from_u), self._p(work), s.stream)

    def ztri_sweep_forward(self, work, edge, c, nchunks):
        _lib.call("ins_slab_ztri_sweep_forward", self._fft, self._p(work), self._p(edge), int(c), int(nchunks), self.setup.stream)

    def ztri_sweep_backward(self, work, edges_all, c, nchunks):
        _lib.call("ins_slab_ztri_sweep_backward", self._fft, self._p(work), self._p(edges_all), int(c), int(nchunks), self.setup.stream)

    def ztri_inverse(self, work, pI):
        _lib.call("ins_slab_ztri_inverse", self._fft, self._p(work), self._p(pI), self.setup.stream)

    def applypressure(self, u, pI, p_top):
        s = self.setup
        _lib.call("ins_slab_applypressure_f64", s.handle, s.ptr(u, True), self._p(pI), self._p(p_top), s.stream)

    def sync(self):
        torch.cuda.synchronize(self.device)


class SlabStepper:
    """Explicit Runge-Kutta stepping of a periodic box decomposed into z-slabs
    (step_explicit_runge_kutta.jl:4-59 distributed; same arithmetic in the same order on every cell)."""

    def __init__(self, method, layout, kernels, comm, chunks=1, zsolve=None):
        self.method, self.lay, self.k, self.comm = method, layout, kernels, comm
        # z direction of the Poisson solve: "tridiag" = distributed tridiagonal systems, one small all-gather per solve;
        # "fft" = two all-to-all transposes around the z-FFT.  Default: tridiag as soon as there is more than one rank
        # (on one rank the fused z-FFT pass is one HBM pass instead of two).
        zsolve = zsolve or os.environ.get("INS_SLAB_ZSOLVE") or ("tridiag" if comm.world > 1 else "fft")
        if zsolve not in ("tridiag", "fft"):
            raise ValueError("zsolve must be 'tridiag' or 'fft'")
        if zsolve == "tridiag" and not bool(getattr(kernels, "supports_ztri", lambda: False)()):
            zsolve = "fft"
        self.zsolve = zsolve
        # kx-chunks of the half spectrum for the pipelined transposes (1 = one all-to-all each way)
        chunks = max(1, min(int(chunks), layout.kxn)) if kernels.can_chunk() else 1
        self.cw = -(-layout.kxn // chunks)  # uniform chunk width (the last chunk may be narrower)
        self.chunks = [(k0, min(self.cw, layout.kxn - k0)) for k0 in range(0, layout.kxn, self.cw)]
        self.packed = bool(getattr(kernels, "is_own", lambda: False)())
        ns = len(method.b)
        self.inkernel = bool(getattr(kernels, "supports_inkernel", lambda: False)()) and len(method.b) > 1
        # Stage-velocity basis (csrc/ins_rk.hip): with in-kernel correction the uncorrected stage velocities V_m stay in memory as the
        # next stencil's input, and V_i = (1 - Σβ) ustart + Σ_{m<i} β_im V_m + Δt A[i,i] k_i with β_i A[0:i,0:i] = A[i,0:i] — no stage force
        # is written or read (RK44: 336 instead of 432 B per cell and step through the stage kernels).
        A = np.asarray(method.A, dtype=float)
        self.vbasis = (self.inkernel and bool(getattr(kernels, "splits_stage", False)) and all(A[i, i] != 0.0 for i in range(ns))
                       and not os.environ.get("INS_RK_KEEP_K"))
        self.wide = bool(getattr(kernels, "wide_stage_kernel", lambda: False)())
        if self.vbasis:
            self.vb = [kernels.vector() for _ in range(ns - 1)]
            self.beta = [np.linalg.solve(A[:i, :i].T, A[i, :i]) if i else np.zeros(0) for i in range(ns)]
            self.ku, self.ub = [], []
        else:
            self.ku = [kernels.vector() for _ in range(ns)]
            self.ub = [kernels.vector(), kernels.vector()]
        plane = layout.n[0] * layout.n[1]
        if self.inkernel:  # pI lives inside the extended buffer so that its ghost planes can be exchanged in place
            self.pX = kernels.pext()
            self.pI = self.pX[plane : plane * (layout.nzl + 1)]
        else:
            self.pX = None
            self.pI = kernels.real()
        self.work = kernels.cplx()
        self.bufa = kernels.cplx()
        self.bufb = kernels.cplx()
        self.p_top = kernels.real()[: layout.n[0] * layout.n[1]].clone()
        if self.zsolve == "tridiag":
            # line ranges of the interface gather (INS_SLAB_ZCHUNKS > 1: the gather of range c travels while range c+1 is swept).  Default 1:
            # with direct sends the gather is ~1 MB over each link (tens of µs), about what the smaller launches cost (+0.08 ms per step at 2 ranges)
            self.zchunks = int(os.environ.get("INS_SLAB_ZCHUNKS") or 1) if bool(getattr(kernels, "ztri_chunked", False)) else 1
            if self.zchunks > 1:
                self.edge_c = [kernels.ztri_chunk_edge(c, self.zchunks) for c in range(self.zchunks)]
                self.edges_all_c = [kernels.ztri_chunk_edge(c, self.zchunks, comm.world) for c in range(self.zchunks)]
            else:
                self.edge = kernels.ztri_edge()
                self.edges_all = kernels.ztri_edge(comm.world)
        self.n = 0

    # -- exchanges ---------------------------------------------------------------------------
    def halo_u(self, u, comps=(0, 1, 2), down_only=False):
        """Fill the z ghost planes of `u`: plane nzl -> next rank's plane 0 (and plane 1 -> prev rank's
        plane nzl+1 unless `down_only`).  Planes span the full padded x/y extent, so edges stay consistent
        (boundary_conditions.jl:97-103)."""
        lay, K = self.lay, self.k
        sends, recvs = [], []
        for c in comps:
            sends.append((K.plane(u, c, lay.nzl), lay.next))
            recvs.append((K.plane(u, c, 0), lay.prev))
        if not down_only:
            for c in comps:
                sends.append((K.plane(u, c, 1), lay.prev))
                recvs.append((K.plane(u, c, lay.nzl + 1), lay.next))
        self.comm.exchange(sends, recvs)

    def halo_u_rest_async(self, u):
        """Every ghost plane of halo_u(u) except the downward w plane (which halo_u(u, comps=(2,), down_only=True) moved
        already), started asynchronously: the stage velocity u* is final once the stencil kernel has run, so these planes can
        travel while the Poisson solve computes."""
        lay, K = self.lay, self.k
        sends = [(K.plane(u, c, lay.nzl), lay.next) for c in (0, 1)] + [(K.plane(u, c, 1), lay.prev) for c in (0, 1, 2)]
        recvs = [(K.plane(u, c, 0), lay.prev) for c in (0, 1)] + [(K.plane(u, c, lay.nzl + 1), lay.next) for c in (0, 1, 2)]
        return self.comm.exchange_async(sends, recvs)

    def halo_p(self):
        """First local pI plane -> previous rank's `p_top`."""
        lay, K = self.lay, self.k
        self.comm.exchange([(K.p_plane(self.pI, 0), lay.prev)], [(self.p_top, lay.next)])

    def halo_p_ext(self, wait=True):
        """Ghost planes of the extended pressure buffer: my last plane -> next rank's `below`; my first two planes ->
        previous rank's two `above` planes."""
        lay, K = self.lay, self.k
        plane = lay.n[0] * lay.n[1]
        pX, nzl = self.pX, lay.nzl
        sends = [(pX[plane * nzl : plane * (nzl + 1)], lay.next), (pX[plane : plane * 3], lay.prev)]
        recvs = [(pX[0:plane], lay.prev), (pX[plane * (nzl + 1) : plane * (nzl + 3)], lay.next)]
        if wait:
            self.comm.exchange(sends, recvs)
            return []
        return self.comm.exchange_async(sends, recvs)

    # -- projection (pressure.jl:69-82 on slabs) ---------------------------------------------------
    def project_(self, u, apply=True, w_halo_done=False, rest_async=False):
        """apply=False: solve only (pI <- p); the gradient-subtract is left to the next stage's stencil kernel."""
        K = self.k
        split = self.zsolve == "tridiag" and self.packed and not w_halo_done and bool(getattr(K, "splits_stage", False)) and self.lay.nzl >= 2
        if split:
            # the w plane below the slab travels while the x pass transforms the planes that do not need it
            lay = self.lay
            pend = self.comm.exchange_async([(K.plane(u, 2, lay.nzl), lay.next)], [(K.plane(u, 2, 0), lay.prev)])
            K.xfwd_planes(u, self.work, 1, lay.nzl - 1)
            for req in pend:
                req.wait()
            if rest_async:  # u is final (apply=False callers): its other ghost planes travel during the rest of the solve
                self._rest = self.halo_u_rest_async(u)
            K.xfwd_planes(u, self.work, 0, 1)
        elif not w_halo_done:
            self.halo_u(u, comps=(2,), down_only=True)  # divergence needs w[I - e_z] only (operators.jl:122)
        if self.zsolve == "tridiag":
            if split:
                src, from_u = u, 2
            elif self.packed:  # power-of-two box: Ω·div(u) formed inside the x pass
                src, from_u = u, 1
            else:
                K.divergence(u, self.pI)
                src, from_u = self.pI, 0
            if self.zchunks > 1:
                nc = self.zchunks
                K.ztri_transform(src, from_u, self.work)
                handles = []
                for c in range(nc):
                    K.ztri_sweep_forward(self.work, self.edge_c[c], c, nc)
                    handles.append(self.comm.all_gather_async(self.edges_all_c[c], self.edge_c[c]))
                for c in range(nc):
                    for req in handles[c]:
                        req.wait()
                    K.ztri_sweep_backward(self.work, self.edges_all_c[c], c, nc)
                K.ztri_inverse(self.work, self.pI)
            else:
                K.ztri_forward(src, from_u, self.work, self.edge)
                self.comm.all_gather(self.edges_all, self.edge)
                K.ztri_finish(self.work, self.edges_all, self.pI)
            if apply:
                self.halo_p()
                K.applypressure(u, self.pI, self.p_top)
            return u
        if self.packed:
            # power-of-two box: [K2 + x pass] -> y pass straight into the send buffer -> chunked transposes around the
            # z pass -> y pass straight out of the receive buffer -> x pass: no divergence / pack / unpack passes
            lay = self.lay
            per_kx = 2 * lay.world * lay.nzl * lay.nyl
            sl = [slice(per_kx * k0, per_kx * (k0 + kc)) for k0, kc in self.chunks]
            K.fft_forward_packed(u, self.work, self.bufa, self.cw)
            fwd = [self.comm.all_to_all_async(self.bufb[r], self.bufa[r], 0) for r in sl]
            bwd = []
            for (k0, kc), r, h in zip(self.chunks, sl, fwd):
                h.wait()
                K.solve_z_chunk(self.bufb[r], k0, kc)
                bwd.append(self.comm.all_to_all_async(self.bufa[r], self.bufb[r], 1))
            for h in bwd:
                h.wait()
            K.fft_inverse_packed(self.bufa, self.work, self.pI, self.cw)
            if apply:
                self.halo_p()
                K.applypressure(u, self.pI, self.p_top)
            return u
        K.divergence(u, self.pI)
        if len(self.chunks) == 1:
            K.fft_forward_xy(self.pI, self.work, self.bufa)
            self.comm.all_to_all(self.bufb, self.bufa)
            K.fft_solve_z(self.bufb)
            self.comm.all_to_all(self.bufa, self.bufb)
            K.fft_inverse_xy(self.bufa, self.work, self.pI)
        else:
            # pipelined over kx-chunks: transpose(c+1) || z-solve(c) || back-transpose(c-1) on two communicators
            lay = self.lay
            per_kx = 2 * lay.world * lay.nzl * lay.nyl  # doubles per unit of kx in a packed chunk
            sl = [slice(per_kx * k0, per_kx * (k0 + kc)) for k0, kc in self.chunks]
            K.fft_xy_forward(self.pI, self.work)
            fwd, bwd = [], []
            for (k0, kc), r in zip(self.chunks, sl):
                K.pack_chunk(self.work, self.bufa[r], k0, kc)
                fwd.append(self.comm.all_to_all_async(self.bufb[r], self.bufa[r], 0))
            for (k0, kc), r, h in zip(self.chunks, sl, fwd):
                h.wait()
                K.solve_z_chunk(self.bufb[r], k0, kc)
                bwd.append(self.comm.all_to_all_async(self.bufa[r], self.bufb[r], 1))
            for (k0, kc), r, h in zip(self.chunks, sl, bwd):
                h.wait()
                K.unpack_chunk(self.bufa[r], self.work, k0, kc)
            K.fft_xy_inverse(self.work, self.pI)
        self.halo_p()
        K.applypressure(u, self.pI, self.p_top)
        return u

    # -- one RK step -------------------------------------------------------------------------
    def step_(self, u, Δt):
        """One RK step; `u` valid (x, y, z ghosts filled on return)."""
        return self._step(u, Δt, False, False)

    def steps_(self, u, Δt, nsteps):
        """`nsteps` steps of size Δt (the fixed-Δt loop of solve_unsteady, solver.jl:74-83).  `u` is valid before and after.  On the fast
        slab pipeline the projection's gradient-subtract of every step but the last is applied in registers by the next step's first
        stage kernel (as csrc/ins_rk.hip does on one GPU): no K4 pass and no full u halo between steps."""
        chain = (self.vbasis and self.wide and bool(getattr(self.k, "splits_stage", False)) and nsteps > 1
                 and not os.environ.get("INS_DISABLE_STEP_CHAIN"))
        if chain and not hasattr(self, "u0"):
            self.u0 = self.k.vector()  # the corrected start field of a chained step
        for n in range(nsteps):
            self._step(u, Δt, chain and n > 0, chain and n < nsteps - 1)
        return u

    def _step(self, u, Δt, raw_in, raw_out):
        """raw_in: `u` holds the previous step's uncorrected result (ghost planes already exchanged / in flight) and pX its pressure;
        raw_out: leave this step's result uncorrected in `u` for the next call."""
        m, K = self.method, self.k
        A = m.A
        ns = len(m.b)
        split = bool(getattr(K, "splits_stage", False))
        if raw_in:
            pending, p_pending = self._carry
        else:
            K.fill_xy_ghosts(u)
            self.halo_u(u)
            pending, p_pending = [], []
        ustart = self.u0 if raw_in else u
        u_in = u
        for i in range(ns):
            coefs, ks, c0m1, self_in = [], [], 0.0, 0.0
            if self.vbasis:
                out = u if i == ns - 1 else self.vb[i]
                for m in range(i):
                    if self.beta[i][m] != 0.0:
                        c0m1 -= float(self.beta[i][m])
                        if m == i - 1 and self.wide:  # V_{i-1} is this stage's stencil input: taken from registers
                            self_in = float(self.beta[i][m])
                            continue
                        coefs.append(float(self.beta[i][m]))
                        ks.append(self.vb[m])
                write_k = False
            else:
                out = u if (i == ns - 1 and ns > 1) else self.ub[i & 1]
                for j in range(i):
                    cf = Δt * A[i, j]
                    if cf != 0.0:
                        coefs.append(cf)
                        ks.append(self.ku[j])
                write_k = any(A[i2, i] != 0.0 for i2 in range(i + 1, ns))
            last = i == ns - 1
            if self.inkernel and (i > 0 or raw_in):
                # previous stage's (or step's) projection applied in registers from (u*, p) — no K4 pass for it.  The planes that read
                # no ghost plane run while the ghost planes of p (and the last of u*) are still arriving.
                kw = self.ku[i] if write_k else None
                us, uso = (None, self.u0) if i == 0 else (ustart, None)  # chained first stage: the corrected input IS the start field
                if split:
                    K.stage_momentum_corr(u_in, self.pX, kw, us, out, coefs, ks, Δt * A[i, i], part=1, c0m1=c0m1, self_in=self_in, ustart_out=uso)
                for req in pending + p_pending:
                    req.wait()
                pending, p_pending = [], []
                if split:
                    K.stage_momentum_corr(u_in, self.pX, kw, us, out, coefs, ks, Δt * A[i, i], part=2, c0m1=c0m1, self_in=self_in, ustart_out=uso)
                else:
                    K.stage_momentum_corr(u_in, self.pX, kw, us, out, coefs, ks, Δt * A[i, i])
            else:
                K.stage_momentum(u_in, self.ku[i] if write_k else None, None if i == 0 else ustart, out, coefs, ks, Δt * A[i, i])
            if self.inkernel and (not last or raw_out):
                # u* is final here (its correction happens inside the next stencil kernel): the w plane the divergence needs
                # goes first, the other five ghost planes travel while the Poisson solve runs
                if split and self.zsolve == "tridiag" and self.packed:
                    # project_ moves the w plane itself beside the x pass and starts the other five planes right behind it
                    self._rest = None
                    self.project_(out, apply=False, rest_async=True)
                    pending = self._rest if self._rest is not None else self.halo_u_rest_async(out)
                else:
                    self.halo_u(out, comps=(2,), down_only=True)
                    pending = self.halo_u_rest_async(out)
                    self.project_(out, apply=False, w_halo_done=True)
                p_pending = self.halo_p_ext(wait=not split)
            else:
                self.project_(out)
                self.halo_u(out)  # z ghost planes for the next stencil (x/y ghosts: K4 images, or periodic addressing in-kernel)
            u_in = out
        self._carry = (pending, p_pending)
        if ns == 1:
            u.copy_(self.ub[0])
        self.n += 1
        return u

    def max_abs_divergence(self, u):
        """Global maximum(abs, Ω⁻¹ pI) diagnostic (uses the projection scratch)."""
        self.halo_u(u, comps=(2,), down_only=True)
        self.k.divergence(u, self.pI)
        v = (self.pI.abs().max() / self.k.cell_volume).reshape(1).to(torch.float64)
        if hasattr(self.comm, "allreduce_"):
            self.comm.allreduce_(v, "max")
        elif self.comm.world > 1:
            vv = v.cpu() if self.comm.backend == "gloo" else v
            dist.all_reduce(vv, op=dist.ReduceOp.MAX, group=self.comm.group)
            v = vv
        return float(v)
