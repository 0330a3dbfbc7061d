#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from two rocprofv3 counter passes of the same command (FETCH_SIZE and WRITE_SIZE, collected
separately as MI355X_MICROARCH.md prescribes; FETCH_SIZE is doubled: gfx950 rule):
    tools/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv out.json "<command>" "<grid>"
Kernels are grouped by a short name derived from the mangled symbol (template arguments kept for the flux kernels)."""
import csv, json, re, sys, collections

def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>(]*>)?", name)  # demangled: (anonymous namespace)::k_flux64<2, 4, true, 1, false>(...)
    if m:
        return m.group(1) + (m.group(2) or "").replace(" ", "")
    return name[:60]

def collect(path, counter):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return acc

fe, wr = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"command": sys.argv[4], "grid": sys.argv[5], "note": "KB counters; hbm_read_GB = 2 x FETCH_SIZE (gfx950), per launch averages", "per_kernel": {}}
for k in sorted(set(fe) | set(wr)):
    f, w = fe.get(k, [0.0]), wr.get(k, [0.0])
    rd, wt = 2 * sum(f) / len(f) * 1024 / 1e9, sum(w) / len(w) * 1024 / 1e9
    if rd + wt < 1e-3:
        continue
    out["per_kernel"][k] = {"launches": len(f), "hbm_read_GB": rd, "hbm_write_GB": wt, "hbm_total_GB": rd + wt}
# stage kernels of the RK loop: FUSE = true instantiations of k_flux64<T, R, XW, FUSE, CORR, SKEL, NW, EXTRA> / k_flux128<T, R, XW, FUSE, CORR, NW>;
# CORR != 0: the correcting form (every stage of a chained run but the very first), CORR == 0: the first stage of a call
def stage_kind(k):
    m = re.match(r"k_flux(?:64|128)<(?:double,)?\d+,\d+,true,(\d)", k)
    return None if not m else ("corr" if m.group(1) != "0" else "first")
for kind in ("corr", "first"):
    sel = [v for k, v in out["per_kernel"].items() if stage_kind(k) == kind]
    if sel:
        n = sum(v["launches"] for v in sel)
        out["per_kernel"][f"stage kernel, {kind}"] = {"hbm_total_GB": sum(v["hbm_total_GB"] * v["launches"] for v in sel) / n, "launches": n}
stage = [v for k, v in out["per_kernel"].items() if stage_kind(k)]
if stage:
    n = sum(v["launches"] for v in stage)
    out["per_kernel"]["stage kernel, RK44 step average"] = {"hbm_total_GB": sum(v["hbm_total_GB"] * v["launches"] for v in stage) / n, "launches": n,
                                                            "note": "launch-weighted over THIS command's launches; bench.py re-weights 'corr' / 'first' for its timed region"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out["per_kernel"].items():
    print(f"{k:50s} {json.dumps(v)}")
