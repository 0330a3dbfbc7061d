#!/usr/bin/env python3
"""Counter table from several `rocprofv3 --pmc` passes of the same command: tools/pmc_table.py "<kernel substring>[,<substring>...]" pass1.csv pass2.csv ...
Rows = dispatches whose kernel name contains one of the substrings, in launch order (the passes run the same launches in the same order); columns = counters."""
import csv, sys, collections, re

keys = sys.argv[1].split(",")
rows = collections.OrderedDict()  # (order index) -> {name, counters}
for path in sys.argv[2:]:
    per = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        kn = r["Kernel_Name"]
        if not any(k in kn for k in keys):
            continue
        did = int(r["Dispatch_Id"])
        d = per.setdefault(did, {"name": kn, "ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for i, (did, d) in enumerate(per.items()):
        row = rows.setdefault(i, {"name": d["name"], "ms": []})
        row["ms"].append(d["ms"])
        for k, v in d.items():
            if k not in ("name", "ms"):
                row[k] = v
def short(n):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>(]*>)?", n)
    return (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else n[:40]
cols = []
for r in rows.values():
    for k in r:
        if k not in ("name", "ms") and k not in cols:
            cols.append(k)
for i, r in rows.items():
    print(f"[{i}] {short(r['name'])}   {min(r['ms']):.4f} ms (min over passes; counters serialise nothing here: {', '.join(f'{m:.3f}' for m in r['ms'])})")
    for c in cols:
        if c in r:
            print(f"      {c:42s} {r[c]:18.0f}")
