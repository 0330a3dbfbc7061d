#!/usr/bin/env python3
"""In-process A/B of the spectral Poisson solve (own FFT passes) under run-time options: tools/zsolve_lab.py N label:OPT=V,... ...
Differences between variants isolate the pass an option acts on (INS_ZSOLVE_TK, INS_ZSOLVE_SKEL, ...)."""
import os, statistics, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
from ins_amd import _lib

n = int(sys.argv[1])
variants = []
for a in sys.argv[2:]:
    label, _, spec = a.partition(":")
    variants.append((label, {k: int(v) for k, _, v in (kv.partition("=") for kv in filter(None, spec.split(",")))}))
allkeys = sorted({k for _, o in variants for k in o})
base = {k: _lib.get_option(k) for k in allkeys}
setup = ins.Setup(x=(np.linspace(0, 1, n + 1),) * 3, Re=1000.0)
ps = ins.psolver_spectral(setup)
p0 = ins.scalarfield(setup); p0.copy_(torch.randn(p0.shape, dtype=torch.float64, device=p0.device))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
times = {l: [] for l, _ in variants}
ref, errs = None, {}
for rep in range(5):
    for label, opts in variants:
        for k in allkeys:
            _lib.set_option(k, opts.get(k, base[k]))
        p = ins.copyfield(p0); ps(p)
        if rep == 0:
            if ref is None: ref = p.clone()
            errs[label] = float((p - ref).abs().max() / ref.abs().max())
        e0.record()
        for _ in range(5): ps(p)
        e1.record(); torch.cuda.synchronize(); times[label].append(e0.elapsed_time(e1) / 5)
for label, ts in times.items():
    print(f"n={n} {label:28s} solve best {min(ts):.4f} ms  median {statistics.median(ts):.4f}   diff vs first {errs[label]:.1e}", flush=True)
