#!/usr/bin/env python3
"""Per-kernel summary (count, total ms, avg us, share) of a rocprofv3 rocpd database -> stdout / CSV."""
import sqlite3, sys

db = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
rows = list(c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                      f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
tot = sum(r[2] for r in rows)
lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
for r in rows:
    lines.append(f'"{r[0]}",{r[1]},{r[2]},{r[3]:.1f},{100*r[2]/tot:.2f},{r[4]},{r[5]}')
if out:
    open(out, "w").write("\n".join(lines) + "\n")
for r in rows[:30]:
    print(f"{r[2]/1e6:9.2f} ms {100*r[2]/tot:5.1f}%  n={r[1]:5d} avg={r[3]/1e3:9.1f} us  {r[0][:120]}")
print(f"total {tot/1e6:.2f} ms")
