#!/usr/bin/env python3
"""Local cost of ONE rank of an N-rank slab run, without peers: a loopback communicator returns the rank's own planes / buffers (wrong
physics across the slab boundary, same kernels and same bytes).  tools/slab_local.py nx ny nz world"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
from bench_dist import tgv_local


class Loopback:
    backend = "loopback"

    def __init__(self, world):
        self.world, self.rank = world, 0

    def exchange(self, sends, recvs):
        for (s, _), (r, _) in zip(sends, recvs):
            r.copy_(s)

    def exchange_async(self, sends, recvs):
        self.exchange(sends, recvs)
        return []

    def all_gather(self, out, inp):
        out.view(self.world, -1).copy_(inp.view(1, -1).expand(self.world, -1))

    def all_gather_async(self, out, inp):
        self.all_gather(out, inp)
        return []


nx, ny, nz, world = (int(a) for a in sys.argv[1:5])
lay = ins.SlabLayout((nx, ny, nz), world, 0)
K = ins.HipSlabKernels(lay, Re=1000.0)
st = ins.SlabStepper(ins.RKMethods.RK44(), lay, K, Loopback(world), zsolve="tridiag")
u = K.vector(); u.copy_(torch.from_numpy(np.ascontiguousarray(tgv_local(lay))).to(u.device))
st.steps_(u, 1e-3, 3)
torch.cuda.synchronize(); t0 = time.perf_counter()
st.steps_(u, 1e-3, 10)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
cells = nx * ny * lay.nzl
print(f"rank-local {nx}x{ny}x{nz}/{world} ({cells/1e6:.1f} M cells/rank): {dt*1e3:.3f} ms/step, {cells/dt/1e6:.0f} M cells/s per rank, finite {bool(torch.isfinite(u).all())}", flush=True)
