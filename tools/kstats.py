"""Per-kernel totals from a rocprofv3 rocpd database: python tools/kstats.py db [steps]"""
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(c.execute("select name, count(*), sum(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot/1e6/steps:.2f} ms/step")
for n, k, t in rows[:40]:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    print(f"{t/1e6/steps:8.3f} ms {k/steps:7.1f}x {t/k/1e3:9.1f} us  {n[:110]}")
