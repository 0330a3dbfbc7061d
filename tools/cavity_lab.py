#!/usr/bin/env python3
"""Config 5 (LidDrivenCavity3D-shaped: cosine x/y walls, periodic z, psolver_direct) step time under run-time options:
tools/cavity_lab.py N label:OPT=V,... ..."""
import os, statistics, sys, time
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
x = (ins.cosine_grid(0.0, 1.0, n), ins.cosine_grid(0.0, 1.0, n), np.linspace(-0.2, 0.2, n + 1))
D, P = ins.DirichletBC, ins.PeriodicBC
setup = ins.Setup(x=x, Re=1000.0, boundary_conditions=((D(), D()), (D(), D((1.0, 0.2, 0.0))), (P(), P())))
ps = ins.psolver_direct(setup)
u0 = ins.velocityfield(setup, lambda a, x, y, z: 0 * (x + y + z), 0.0, psolver=ps, doproject=False)
m = ins.RKMethods.RK44()
cache = ins.ode_method_cache(m, setup, ps)
h = 0.9 * ins.get_cfl_timestep_(None, u0, setup)
times = {l: [] for l, _ in variants}
ref, errs = None, {}
for rep in range(4):
    for label, opts in variants:
        for k in allkeys:
            _lib.set_option(k, opts.get(k, base[k]))
        st = ins.create_stepper(m, setup=setup, psolver=ps, u=ins.copyfield(u0), t=0.0)
        for _ in range(2):
            st = ins.timestep_(m, st, h, cache=cache)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            st = ins.timestep_(m, st, h, cache=cache)
        torch.cuda.synchronize(); times[label].append((time.perf_counter() - t0) / 5 * 1e3)
        if rep == 0:
            if ref is None: ref = st.u.clone()
            errs[label] = float(torch.nan_to_num(st.u - ref).abs().max() / torch.nan_to_num(ref).abs().max())  # never-written slots behind walls hold garbage
for label, ts in times.items():
    print(f"cavity n={n} {label:24s} best {min(ts):.3f} ms/step  median {statistics.median(ts):.3f}  diff vs first {errs[label]:.1e}", flush=True)
