#!/usr/bin/env python3
"""Time one spectral solve (psolver(p), all passes) on an nx x ny x nz periodic box under option sets; the solver is created under the options:
tools/solve_lab.py NX NY NZ label:OPT=V,...   (e.g. own: rocfft:INS_OWNFFT_POW2_ONLY=1)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
from ins_amd import _lib
n = tuple(int(v) for v in sys.argv[1:4])
sp = ins.Setup(x=tuple(np.linspace(0.0, 1.0, m + 1) for m in n), Re=1000.0)
p = ins.scalarfield(sp)
p.copy_(torch.randn(p.shape, dtype=torch.float64, device=p.device))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ref = None
for a in sys.argv[4:]:
    label, _, spec = a.partition(":")
    opts = {k: int(v) for k, _, v in (kv.partition("=") for kv in filter(None, spec.split(",")))}
    with _lib.options(**opts):
        ps = ins.psolver_spectral(sp)
        q = ins.copyfield(p)
        ins.poisson_(ps, q)
        if ref is None:
            ref = q.clone()
        err = float((q - ref).abs().max() / ref.abs().max())
        best = 1e9
        for _ in range(5):
            e0.record()
            for _ in range(10):
                ins.poisson_(ps, q)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
    print(f"n={n} {label:24s} {best*1e3:9.1f} us per solve   diff vs first {err:.1e}", flush=True)
    del ps
