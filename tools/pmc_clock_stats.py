"""Reduce a `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES` run of tools/pmc_clock_target.py (rocpd sqlite):
per dispatch, duration from the kernel trace and the two counters -> effective clock = GRBM_GUI_ACTIVE / duration."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cc = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
kc = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("# counters_collection columns:", cc)
print("# kernels columns:", kc)
val = collections.defaultdict(dict)
name_of = {}
nrows = {}
for did, kn, cn, v, n in db.execute("select dispatch_id, kernel_name, counter_name, sum(value), count(*) from counters_collection "
                                    "group by dispatch_id, counter_name"):
    val[did][cn] = v / n  # one row per counter instance (XCC / SE dimension): the MEAN over instances is a per-instance cycle count
    nrows[cn] = n
    name_of[did] = kn
print("# rows (instances) per dispatch and counter:", nrows, "-> values below are means over instances")
dur = {}
if "dispatch_id" in kc:
    for did, st, en in db.execute("select dispatch_id, start, end from kernels"):
        dur[did] = en - st
elif "dispatch_id" in cc and "start" in cc:
    for did, st, en in db.execute("select dispatch_id, min(start), max(end) from counters_collection group by dispatch_id"):
        dur[did] = en - st
rows = []
for did in sorted(val):
    if "gemm_bf16" not in name_of[did]:
        continue
    d = dur.get(did)
    gui, sq = val[did].get("GRBM_GUI_ACTIVE"), val[did].get("SQ_BUSY_CYCLES")
    rows.append((did, name_of[did][:48], d, gui, sq))
print(f"{'dispatch':>8s} {'kernel':48s} {'dur_us':>9s} {'GRBM_GUI_ACTIVE':>16s} {'SQ_BUSY_CYCLES':>15s} {'GUI/dur GHz':>12s}")
for did, kn, d, gui, sq in rows:
    ghz = (gui / d) if (d and gui) else float("nan")
    print(f"{did:8d} {kn:48s} {(d or 0) / 1e3:9.1f} {gui or 0:16.0f} {sq or 0:15.0f} {ghz:12.3f}")
