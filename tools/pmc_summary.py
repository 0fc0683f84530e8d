#!/usr/bin/env python
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database (one --pmc pass)."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [d[1] for d in cur.execute("pragma table_info('counters_collection')")]
rows = cur.execute("select * from counters_collection").fetchall()
ix = {c: i for i, c in enumerate(cols)}
name_c = "kernel_name" if "kernel_name" in ix else ("name" if "name" in ix else None)
cn = "counter_name" if "counter_name" in ix else None
val = "value" if "value" in ix else ("counter_value" if "counter_value" in ix else None)
if not (name_c and cn and val):
    print("columns:", cols); print(rows[:3]); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r[ix[name_c]]][r[ix[cn]]].append(float(r[ix[val]]))
ctrs = sorted({c for k in agg.values() for c in k})
NAME_W = 160      # full template argument lists (at 60 columns two instantiations of one kernel template shared a row)
print(f"{'kernel':{NAME_W}s} {'n':>4s} " + " ".join(f"{c:>22s}" for c in ctrs))
for k, d in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    n = max(len(v) for v in d.values())
    print(f"{k[:NAME_W]:{NAME_W}s} {n:4d} " + " ".join(f"{(sum(d[c]) / len(d[c]) if c in d else 0):22.4g}" for c in ctrs))
