"""Per-stream busy time of a rocprofv3 kernel trace: span, sum of kernel durations, union, big gaps."""
import collections
import glob
import sqlite3
import sys

db = sqlite3.connect(glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0])
rows = list(db.execute("select name,start,end,stream_id from kernels order by start"))
t0, t1 = rows[0][1], max(r[2] for r in rows)
print(f"trace span {(t1 - t0) / 1e6:.1f} ms, {len(rows)} kernels")
by = collections.defaultdict(list)
for n, s, e, st in rows:
    by[st].append((s, e, n))
for st, ks in sorted(by.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _ in ks)
    span = ks[-1][1] - ks[0][0]
    union, cur_s, cur_e = 0, ks[0][0], ks[0][1]
    gaps = []
    for (s, e, n), prev in zip(ks[1:], ks):
        if s > cur_e:
            union += cur_e - cur_s
            if s - cur_e > 1e6:
                gaps.append(((cur_e - t0) / 1e6, (s - cur_e) / 1e6, prev[2][:50], n[:50]))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    print(f"stream {st}: {len(ks)} kernels, first at {(ks[0][0]-t0)/1e6:.1f} ms, span {span/1e6:.1f} ms, "
          f"sum {busy/1e6:.1f} ms, union {union/1e6:.1f} ms, idle {(span-union)/1e6:.1f} ms")
    for at, g, a, b in gaps[:40]:
        print(f"    gap {g:7.1f} ms at {at:8.1f} ms  after [{a}] before [{b}]")
