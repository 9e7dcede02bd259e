"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace: per-kernel count / total / avg.
    python tools/rocpd_stats.py gpurun_out/prof_bench/bench_results.db > profiles/xyz.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute(
    "select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
    "max(end-start)/1e3 from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"# source: {sys.argv[1]}  (rocprofv3 --kernel-trace --stats), total kernel time {tot/1e3:.2f} ms")
print(f"{'kernel':78s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for name, n, total, avg, mn, mx in rows:
    if total / tot < 0.0005:
        continue
    print(f"{name[:78]:78s} {n:7d} {total/1e3:10.3f} {avg:10.2f} {mn:9.2f} {mx:9.2f} {100*total/tot:6.2f}")
