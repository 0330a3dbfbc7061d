#!/usr/bin/env python3
"""Run one step-adjacent operator a few times (for rocprofv3 --pmc): tools/fields_one.py name [n]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
name = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
setup = ins.Setup(x=(np.linspace(0, 1, n + 1),) * 3)
u = ins.vectorfield(setup); u.copy_(torch.randn(u.shape, dtype=torch.float64, device=u.device))
q = ins.scalarfield(setup); sig = ins.tensorfield(setup); F = ins.vectorfield(setup)
fn = dict(strain=lambda: ins.dissipation_from_strain_(q, u, setup), eig2=lambda: ins.eig2field_(q, u, setup), smag=lambda: ins.smagtensor_(sig, u, 0.1, setup),
          div=lambda: ins.divoftensor_(F, sig, setup), vort=lambda: ins.vorticity_(F, u, setup), q=lambda: ins.Qfield_(q, u, setup))[name]
for _ in range(3): fn()
torch.cuda.synchronize()
