#!/usr/bin/env python3
"""In-process A/B of the whole RK44 step (chained `timesteps_`) under run-time options, one allocation:

    tools/step_lab.py N label:OPT=V,OPT=V ...

TGV3D N^3 (dt 1e-3 at 256, 2.5e-4 at 512); every variant is timed round-robin (best / median of 5 x K steps) and checked
against the first variant's result (relative max-norm after the same number of steps from the same start)."""
import os
import statistics
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins  # noqa: E402
from ins_amd import _lib  # noqa: E402


def tgv3d(al, x, y, z):
    if al == 0:
        return np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y) * np.sin(2 * np.pi * z) / 2
    if al == 1:
        return -np.cos(2 * np.pi * x) * np.sin(2 * np.pi * y) * np.sin(2 * np.pi * z) / 2
    return 0 * (x + y + z)


def main():  # noqa: C901
    n = int(sys.argv[1])
    variants = []
    for a in sys.argv[2:]:
        label, _, spec = a.partition(":")
        opts = {}
        for kv in filter(None, spec.split(",")):
            k, _, v = kv.partition("=")
            opts[k] = int(v)
        variants.append((label, opts))
    allkeys = sorted({k for _, o in variants for k in o})
    base = {k: _lib.get_option(k) for k in allkeys}

    def apply(opts):
        for k in allkeys:
            _lib.set_option(k, opts.get(k, base[k]))

    setup = ins.Setup(x=(np.linspace(0.0, 1.0, n + 1),) * 3, Re=1000.0)
    ps = ins.psolver_spectral(setup)
    u0 = ins.velocityfield(setup, tgv3d, 0.0, psolver=ps)
    m = ins.RKMethods.RK44()
    cache = ins.ode_method_cache(m, setup, ps)
    dt = 1e-3 if n <= 256 else 2.5e-4
    K = 10 if n <= 256 else 4
    times = {label: [] for label, _ in variants}
    ref, errs = None, {}
    for rep in range(5):
        for label, opts in variants:
            apply(opts)
            st = ins.create_stepper(m, setup=setup, psolver=ps, u=ins.copyfield(u0), t=0.0)
            st = ins.timesteps_(m, st, dt, 2, cache=cache)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = ins.timesteps_(m, st, dt, K, cache=cache)
            torch.cuda.synchronize()
            times[label].append((time.perf_counter() - t0) * 1e3 / K)
            if rep == 0:
                if ref is None:
                    ref = st.u.clone()
                errs[label] = float((st.u - ref).abs().max() / ref.abs().max())
    for label, ts in times.items():
        b = min(ts)
        print(f"n={n} {label:32s} best {b:.4f} ms/step  median {statistics.median(ts):.4f}   {n**3 / b / 1e3:7.0f} Mcell/s   diff vs first {errs[label]:.1e}", flush=True)


if __name__ == "__main__":
    main()
