"""Per-position kernel durations inside a fit step from a rocprofv3 kernel trace (fit only)."""
import collections
import glob
import sqlite3
import sys

db = sqlite3.connect(glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0])
rows = list(db.execute("select name,start,end from kernels order by start"))
steps, cur = [], []
for n, s, e in rows:
    short = n.replace("(anonymous namespace)::", "").replace("void ", "").split('(')[0][:48]
    if short.startswith("fit_prep") or short.startswith("fit_rows_kernel"):
        if cur:
            steps.append(cur)
        cur = []
    cur.append((short, s, e))
steps.append(cur)
for length in sorted({len(st) for st in steps}):
    sel = [st for st in steps if len(st) == length and (st[0][0].startswith("fit_prep") or st[0][0].startswith("fit_rows"))]
    if len(sel) < 50:
        continue
    print(f"--- steps with {length} kernels: {len(sel)}")
    tot = 0.0
    for i in range(length):
        d = sum(st[i][2] - st[i][1] for st in sel) / len(sel) / 1e3
        gap = sum(st[i][1] - st[i - 1][2] for st in sel) / len(sel) / 1e3 if i else 0.0
        tot += d + gap
        print(f"  {i:2d} {sel[0][i][0]:48s} {d:7.2f} us   gap before {gap:5.2f}")
    print(f"  step total {tot:.1f} us")
