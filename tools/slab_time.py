#!/usr/bin/env python3
"""One-rank timing of the slab pipeline (what every rank does locally in the multi-GPU run, without the exchanges):
tools/slab_time.py [n]  — RK44 step at n^3 with zsolve = fft (fused z-FFT pass) and zsolve = tridiag (csrc/ins_ztri.hip)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_dist import tgv_local

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for zs in ("fft", "tridiag"):
    lay = ins.SlabLayout((n, n, n), 1, 0)
    K = ins.HipSlabKernels(lay, Re=1000.0)
    st = ins.SlabStepper(ins.RKMethods.RK44(), lay, K, ins.SlabComm(), chunks=4, zsolve=zs)
    u = K.vector(); u.copy_(torch.from_numpy(np.ascontiguousarray(tgv_local(lay))).to(u.device))
    st.project_(u); st.halo_u(u)
    st.steps_(u, 1e-3, 3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st.steps_(u, 1e-3, 10)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"slab 1 rank {n}^3 zsolve={zs}: {dt*1e3:.3f} ms/step, {n**3/dt/1e6:.0f} M cells/s, div*dx {st.max_abs_divergence(u)/n:.2e}", flush=True)
    del st, K
