"""Attribute start delays of fit-stream kernels to the ViT kernel running at that moment."""
import bisect
import collections
import glob
import sqlite3
import sys

db = sqlite3.connect(glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0])
rows = list(db.execute("select name,start,end,stream_id from kernels order by start"))
by = collections.defaultdict(list)
for n, s, e, st in rows:
    by[st].append((s, e, n))
streams = sorted(by, key=lambda k: -len(by[k]))
fit, others = by[streams[0]], [by[s] for s in streams[1:]]
vit = max(others, key=lambda iv: sum(e - s for s, e, _ in iv))
vs = [s for s, e, n in vit]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
idle = [0, 0.0]
for (s0, e0, n0), (s1, e1, n1) in zip(fit, fit[1:]):
    gap = (s1 - e0) / 1e3
    if gap > 2000:  # image boundary
        continue
    i = bisect.bisect_right(vs, s1) - 1
    if i >= 0 and vit[i][1] > e0:  # a ViT kernel overlaps the gap
        key = vit[i][2].split('(')[0][-40:]
        a = agg[key]
        a[0] += 1; a[1] += gap; a[2] += (e1 - s1) / 1e3
    else:
        idle[0] += 1; idle[1] += gap
print(f"fit stream {streams[0]}: {len(fit)} kernels; gaps while no ViT kernel runs: n={idle[0]} mean {idle[1]/max(idle[0],1):.2f} us")
for k, (n, g, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  behind {k:42s} n={n:6d} mean gap {g/n:7.2f} us  (mean fit kernel {d/n:6.2f} us)  total gap {g/1e3:8.1f} ms")
