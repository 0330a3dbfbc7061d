#!/usr/bin/env python3
"""Are small periodic boxes launch-bound?  ms/step through timesteps_ (one native call for K steps), the step replayed as a hipGraph against the plain
launch loop (INS_DISABLE_STEP_GRAPH=1): tools/small_grid_lab.py n ...   (LAB_2D=1: n x n grids)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
from ins_amd import _lib

D = 2 if os.environ.get("LAB_2D") else 3
K = 50
for n in [int(a) for a in sys.argv[1:]]:
    for label, opts in (("warm-up", {}), ("plain", {"INS_DISABLE_STEP_GRAPH": 1}), ("separate x / y passes", {"INS_DISABLE_XYFUSED": 1}), ("graph", {"INS_STEP_GRAPH": 1})):
        with _lib.options(**opts):
            sp = ins.Setup(x=(np.linspace(0, 1, n + 1),) * D, Re=1000.0)
            ps = ins.psolver_spectral(sp)
            u = ins.random_field(sp, kp=4, seed=0, psolver=ps)
            m = ins.RKMethods.RK44()
            cache = ins.ode_method_cache(m, sp, ps)
            st = ins.create_stepper(m, setup=sp, psolver=ps, u=u, t=0.0)
            st = ins.timesteps_(m, st, 1e-4, K, cache=cache)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            st = ins.timesteps_(m, st, 1e-4, K, cache=cache)
            e1.record()
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            print(f"n={n}^{D} {label}: wall {t_all/K*1e3:.3f} ms/step, GPU span {e0.elapsed_time(e1)/K:.3f} ms/step, host issue {t_issue/K*1e3:.3f} ms/step", flush=True)
            del cache, st, ps, sp
