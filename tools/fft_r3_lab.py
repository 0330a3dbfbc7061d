"""Step time on 192^3 / 384^3 boxes: own radix-3-fronted passes vs rocFFT plans (INS_OWNFFT_POW2_ONLY=1)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins  # noqa: E402
from ins_amd import _lib  # noqa: E402

for n in (192, 384):
    for pow2_only in (0, 1):
        _lib.set_option("INS_OWNFFT_POW2_ONLY", pow2_only)
        sp = ins.Setup(x=(np.linspace(0, 1, n + 1),) * 3, Re=2000.0)
        ps = ins.psolver_spectral(sp)
        u = ins.random_field(sp, kp=10, A=1.0, seed=0, psolver=ps)
        m = ins.RKMethods.RK44()
        cache = ins.ode_method_cache(m, sp, ps)
        st = ins.create_stepper(m, setup=sp, psolver=ps, u=u, t=0.0)
        st = ins.timesteps_(m, st, 1e-4, 5, cache=cache)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 20
        st = ins.timesteps_(m, st, 1e-4, k, cache=cache)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k
        print(f"n={n} pow2_only={pow2_only}: {dt*1e3:.3f} ms/step  div={ins.max_abs_divergence(st.u, sp):.2e}", flush=True)
        del cache, st, ps, sp, u
