#!/usr/bin/env python3
"""Closure-force lab: event time of smagorinsky_closure(setup)(u, θ) at n³ periodic as one kernel (csrc/ins_smagforce.hip) for several z-chunks, and
as the reference's three kernels: tools/smagforce_lab.py n"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
from ins_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = tuple(np.linspace(0, 1, n + 1) for _ in range(3))
setup = ins.Setup(x=x, Re=1000.0)
u = ins.random_field(setup, 0.0)
m = ins.smagorinsky_closure(setup)
def t(label, **opts):
    with _lib.options(**opts):
        for _ in range(3): m(u, 0.17)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): m(u, 0.17)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
    print(f"{label}: {ms:.3f} ms  ({6 * 8 * n**3 / ms / 1e9:.2f} TB/s of 3 in + 3 out)", flush=True)
t("three kernels", INS_DISABLE_SMAGFORCE=1)
for zc in (16, 32, 64):
    t(f"one kernel zc={zc}", INS_SMAGFORCE_ZC=zc)
t("one kernel, barrier per plane", INS_SMAGFORCE_BAR=1)
t("generalised form, uniform", INS_SMAGFORCE_FORCE_GEN=1)
t("generalised form, metric tables", INS_SMAGFORCE_FORCE_GEN=2)
