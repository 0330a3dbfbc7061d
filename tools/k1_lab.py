#!/usr/bin/env python3
"""In-process A/B of K1 variants on ONE allocation through the run-time options (ins_set_option):

    tools/k1_lab.py N [--once] label:OPT=V,OPT=V ...

Plain momentum kernel at N^3 on random data; every variant is timed round-robin (best / median of 6 x 5 launches).
--once: one launch per variant in the given order (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; parse with
tools/k1_lab.py parse N fetch.csv write.csv label ...)."""
import csv
import os
import statistics
import sys


def parse_variants(args):
    out = []
    for a in args:
        label, _, spec = a.partition(":")
        opts = {}
        for kv in filter(None, spec.split(",")):
            k, _, v = kv.partition("=")
            opts[k] = int(v)
        out.append((label, opts))
    return out


def run(n, variants, once):
    import numpy as np
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import ins_amd as ins
    from ins_amd import _lib

    setup = ins.Setup(x=(np.linspace(0, 1, n + 1),) * 3, Re=1000.0)
    torch.manual_seed(0)
    u = ins.vectorfield(setup)
    u.copy_(torch.randn(u.shape, dtype=torch.float64, device=u.device))
    ins.apply_bc_u_(u, 0.0, setup)
    F = ins.vectorfield(setup)
    allkeys = sorted({k for _, o in variants for k in o})
    base = {k: _lib.get_option(k) for k in allkeys}

    def apply(opts):
        for k in allkeys:
            _lib.set_option(k, opts.get(k, base[k]))

    if once:
        for label, opts in variants:
            apply(opts)
            ins.momentum_(F, u, None, 0.0, setup)
            torch.cuda.synchronize()
        F.copy_(u)  # the flat copy of the same arrays, for the counter passes
        torch.cuda.synchronize()
        return
    # reference result for a correctness check of every variant
    apply({"INS_DISABLE_FLUX64": 1})
    ref = ins.momentum(u, None, 0.0, setup)
    scale = float(ref.abs().max())
    times = {label: [] for label, _ in variants}
    errs = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(6):
        for label, opts in variants:
            apply(opts)
            ins.momentum_(F, u, None, 0.0, setup)
            if rep == 0:
                errs[label] = float((F - ref).abs().max()) / scale
            e0.record()
            for _ in range(5):
                ins.momentum_(F, u, None, 0.0, setup)
            e1.record()
            torch.cuda.synchronize()
            times[label].append(e0.elapsed_time(e1) / 5)
    # the box's flat-copy rate on the same two arrays (same bytes as K1's compulsory traffic)
    cp = []
    for rep in range(4):
        e0.record()
        for _ in range(5):
            F.copy_(u)
        e1.record()
        torch.cuda.synchronize()
        cp.append(e0.elapsed_time(e1) / 5)
    print(f"n={n} {'torch copy_ (same arrays)':28s} best {min(cp):.4f} ms  {48.0 * n**3 / min(cp) / 1e6:6.0f} GB/s", flush=True)
    for label, ts in times.items():
        b = min(ts)
        print(f"n={n} {label:28s} best {b:.4f} ms  median {statistics.median(ts):.4f} ms  {48.0 * n**3 / b / 1e6:6.0f} GB/s = {48.0 * n**3 / b / 1e6 / 8000:.3f} of 8 TB/s"
              f"   err {errs[label]:.1e}", flush=True)


def parse(n, fetch_csv, write_csv, labels):
    def vals(path, name):
        out = []
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == name and ("k_flux64" in row["Kernel_Name"] or "k_momentum_flux" in row["Kernel_Name"]):
                out.append((float(row["Counter_Value"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6))
        return out

    fe, wr = vals(fetch_csv, "FETCH_SIZE"), vals(write_csv, "WRITE_SIZE")
    alg = 24.0 * n**3 / 1e9
    print(f"n={n}: algorithmic read = write = {alg:.3f} GB   (FETCH_SIZE KB x2: gfx950 rule; WRITE_SIZE KB)")
    for label, (f, t1), (w, t2) in zip(labels, fe, wr):
        rd, wt = 2 * f * 1024 / 1e9, w * 1024 / 1e9
        t = min(t1, t2)
        print(f"  {label:28s} read {rd:.3f} GB ({rd / alg:.2f}x)  write {wt:.3f} GB ({wt / alg:.2f}x)  {t:.4f} ms  L2-egress {(rd + wt) / t:.2f} TB/s  algorithmic {2 * alg / t:.2f} TB/s")


if __name__ == "__main__":
    if sys.argv[1] == "parse":
        parse(int(sys.argv[2]), sys.argv[3], sys.argv[4], [a.partition(":")[0] for a in sys.argv[5:]])
    else:
        once = "--once" in sys.argv
        args = [a for a in sys.argv[2:] if a != "--once"]
        run(int(sys.argv[1]), parse_variants(args), once)
