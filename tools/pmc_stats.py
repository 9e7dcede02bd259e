"""Summarise rocprofv3 --pmc results (rocpd sqlite): per kernel name, mean counter values."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
print("# counters_collection columns:", cols)
rows = db.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection "
                  "group by kernel_name, counter_name").fetchall() if "kernel_name" in cols else []
agg = collections.defaultdict(dict)
for k, c, v, n in rows:
    agg[k][c] = (v, n)
for k, d in sorted(agg.items(), key=lambda kv: -max(v for v, _ in kv[1].values())):
    n = max(n for _, n in d.values())
    print(f"{k[:90]}  dispatches={n}")
    for c, (v, nn) in sorted(d.items()):
        print(f"    {c:32s} total={v:.6g}  per-dispatch={v / max(nn, 1):.6g}")
