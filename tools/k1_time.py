#!/usr/bin/env python3
"""Time the plain momentum kernel (K1) at n^3 for the configuration given by INS_FLUX_* env knobs."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
for n in [int(a) for a in sys.argv[1:]] or [512]:
    setup = ins.Setup(x=(np.linspace(0, 1, n + 1),) * 3, Re=1000.0)
    torch.manual_seed(0)
    u = ins.vectorfield(setup); u.copy_(torch.randn(u.shape, dtype=torch.float64, device=u.device)); ins.apply_bc_u_(u, 0.0, setup)
    F = ins.vectorfield(setup)
    for _ in range(3): ins.momentum_(F, u, None, 0.0, setup)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(3):
        e0.record()
        for _ in range(10): ins.momentum_(F, u, None, 0.0, setup)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print(f"n={n} rows={os.environ.get('INS_FLUX_ROWS','-')} zc={os.environ.get('INS_FLUX_ZC','-')} xw={os.environ.get('INS_FLUX_XW','-')}: {best:.4f} ms {48.0*n**3/best/1e6:.0f} GB/s", flush=True)
