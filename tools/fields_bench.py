#!/usr/bin/env python3
"""Bandwidth of the step-adjacent operators at n^3, periodic: tools/fields_bench.py [n]
Algorithmic bytes per cell = compulsory reads + writes of the fields each entry point touches (stated per row)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bcT = ((ins.PeriodicBC(), ins.PeriodicBC()),) * 3
T = ins.temperature_equation(Pr=0.71, Ra=1e6, Ge=0.1, boundary_conditions=bcT)
setup = ins.Setup(x=(np.linspace(0, 1, n + 1),) * 3, temperature=T)
g = torch.Generator(device=setup.device).manual_seed(0)
def rnd(f):
    f.copy_(torch.randn(f.shape, dtype=torch.float64, device=f.device, generator=g)); return f
u, w, up = rnd(ins.vectorfield(setup)), ins.vectorfield(setup), ins.vectorfield(setup)
p, q, temp, c = rnd(ins.scalarfield(setup)), ins.scalarfield(setup), rnd(ins.scalarfield(setup)), ins.scalarfield(setup)
G, diff, F = ins.vectorfield(setup), ins.vectorfield(setup), ins.vectorfield(setup)
sig = ins.tensorfield(setup)
spec = ins.processors._Spectrum(setup, ins.spectral_stuff(setup)["inds"])
rows = [
    ("vorticity", 48, lambda: ins.vorticity_(w, u, setup)),
    ("interpolate_u_p", 48, lambda: ins.interpolate_u_p_(up, u, setup)),
    ("interpolate_w_p", 48, lambda: ins.interpolate_ω_p_(up, w, setup)),
    ("Qfield", 32, lambda: ins.Qfield_(q, u, setup)),
    ("dissipation_from_strain", 32, lambda: ins.dissipation_from_strain_(q, u, setup)),
    ("eig2field", 32, lambda: ins.eig2field_(q, u, setup)),
    ("Dfield (+pressuregradient)", 64, lambda: ins.Dfield_(q, G, p, setup)),
    ("apply_bc_temp", 0, lambda: ins.apply_bc_temp_(temp, 0.0, setup)),
    ("convection_diffusion_temp", 48, lambda: ins.convection_diffusion_temp_(c, u, temp, setup)),
    # algorithmic bytes: read u (24), write diss (8) and the `diff` argument the reference leaves filled with diffusion(u) (24); the three-pass form as built
    # moves 184 B per cell (fill 24 + diffusion 24 + 24 + interpolation 24 + 24 + 8 and re-reads), which is what round 2 credited it with
    ("dissipation (fill + diffusion + interp)", 56, lambda: ins.dissipation_(c, diff, u, setup)),
    ("gravity", 24, lambda: ins.gravity_(F, temp, setup)),
    ("smagtensor", 72, lambda: ins.smagtensor_(sig, u, 0.1, setup)),
    ("divoftensor", 72, lambda: ins.divoftensor_(F, sig, setup)),
    # priced as round 2 did (per component: the real array in, the half spectrum out); what must move at least is u in: 24 B per cell; as built it is three
    # passes per component (x incl. the ghost strip 16 B, y 16 B, z 16 B per cell) + the shell gather
    ("spectrum (3 x [x + y + z pass] + shells)", 3 * (16 + 8), lambda: spec(u)),
]
out = {}
for name, bpc, fn in rows:
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gbs = bpc * n**3 / ms / 1e6
    out[name] = dict(ms=round(ms, 4), bytes_per_cell=bpc, GBs=round(gbs, 1), frac_of_8TBs=round(gbs / 8000, 3))
    print(f"{name:42s} {ms:8.4f} ms  {bpc:4d} B/cell  {gbs:8.1f} GB/s  ({gbs/80:.1f} % of 8 TB/s)", flush=True)
print(json.dumps(dict(n=n, ops=out)))
