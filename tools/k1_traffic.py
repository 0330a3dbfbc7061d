#!/usr/bin/env python3
"""HBM traffic of K1 variants, one launch each, in a fixed order (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
passes, --output-format csv).  `tools/k1_traffic.py n` runs; `tools/k1_traffic.py parse n fetch.csv write.csv` tabulates."""
import os, sys, csv

# (label, disable_flux64, rows, zc, lds)
VARIANTS = ([(f"old62 R{r} zc{zc}", 1, r, zc, 0) for r in (2, 4) for zc in (8, 16, 32, 64)]
            + [(f"f64 R{r} zc{zc} lds{lds//1000}k", 0, r, zc, lds) for r in (2, 4) for zc in (8, 32, 64) for lds in (0, 70000)])

def run(n):
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import ins_amd as ins
    from ins_amd import _lib
    tune = _lib.load().ins_tune_flux64
    tune_old = _lib.load().ins_tune_flux3d
    setup = ins.Setup(x=(np.linspace(0, 1, n + 1),) * 3, Re=1000.0)
    torch.manual_seed(0)
    u = ins.vectorfield(setup); u.copy_(torch.randn(u.shape, dtype=torch.float64, device=u.device)); ins.apply_bc_u_(u, 0.0, setup)
    F = ins.vectorfield(setup)
    for (label, dis, r, zc, lds) in VARIANTS:
        tune(dis, r, -1, zc, -1, 0, 0, lds)
        tune_old(r, zc, 0)
        ins.momentum_(F, u, None, 0.0, setup)
        torch.cuda.synchronize()

def parse(n, fetch_csv, write_csv):
    def vals(path, name):
        out = []
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == name and ("k_flux64" in row["Kernel_Name"] or "k_momentum_flux" in row["Kernel_Name"]):
                out.append((float(row["Counter_Value"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6))
        return out
    fe, wr = vals(fetch_csv, "FETCH_SIZE"), vals(write_csv, "WRITE_SIZE")
    alg = 24.0 * n**3 / 1e9
    print(f"n={n}: algorithmic read = write = {alg:.3f} GB   (FETCH_SIZE KB x2: gfx950 rule; WRITE_SIZE KB)")
    for (label, *_), (f, t1), (w, t2) in zip(VARIANTS, fe, wr):
        rd, wt = 2 * f * 1024 / 1e9, w * 1024 / 1e9
        t = min(t1, t2)
        print(f"  {label:22s} read {rd:.3f} GB ({rd/alg:.2f}x)  write {wt:.3f} GB ({wt/alg:.2f}x)  {t:.4f} ms  L2-egress {(rd+wt)/t:.2f} TB/s  algorithmic {2*alg/t:.2f} TB/s")

if __name__ == "__main__":
    if sys.argv[1] == "parse":
        parse(int(sys.argv[2]), sys.argv[3], sys.argv[4])
    else:
        run(int(sys.argv[1]))
