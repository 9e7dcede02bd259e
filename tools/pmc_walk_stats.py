"""Reduce a `rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE` run of
tools/pmc_target_walks.py (rocpd sqlite): the GEMM dispatches in launch order (the demangler truncates all three kernels to one
name; the order is the target's: shapes qkv, fc1, fc2 x 4 rounds x schedules 13, 4, 11; round 0 is dropped from the means), per dispatch the duration from the
kernel trace and every counter (summed over its instances), then the mean of each (shape, schedule) triple."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
kc = [r[1] for r in db.execute("pragma table_info(kernels)")]
val = collections.defaultdict(dict)
name_of = {}
for did, kn, cn, v in db.execute("select dispatch_id, kernel_name, counter_name, sum(value) from counters_collection "
                                 "group by dispatch_id, counter_name"):
    val[did][cn] = v
    name_of[did] = kn
dur = {}
if "dispatch_id" in kc:
    for did, st, en in db.execute("select dispatch_id, start, end from kernels"):
        dur[did] = en - st
else:
    for did, st, en in db.execute("select dispatch_id, min(start), max(end) from counters_collection group by dispatch_id"):
        dur[did] = en - st
ids = [d for d in sorted(val) if "gemm_bf16" in name_of[d]]
names = ["SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"]
shapes, scheds = ["qkv  N=2304 K= 768", "fc1  N=3072 K= 768", "fc2  N= 768 K=3072"], [13, 4, 11]
print(f"{'dispatch':>8s} {'shape':20s} {'sched':>5s} {'dur_us':>9s} " + " ".join(f"{n:>26s}" for n in names))
agg = collections.defaultdict(list)
for i, did in enumerate(ids):
    sh, rnd, sc = shapes[(i // 12) % 3], (i // 3) % 4, scheds[i % 3]  # per shape: 4 rounds x (13, 4, 11)
    row = [dur.get(did, 0) / 1e3] + [val[did].get(n, float("nan")) for n in names]
    if rnd > 0:
        agg[(sh, sc)].append(row)
    print(f"{did:8d} {sh:20s} {sc:5d} {row[0]:9.1f} " + " ".join(f"{x:26.6g}" for x in row[1:]))
print("\nmeans per (shape, schedule); MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CYCLES... see profiles/r06/README.md)")
for (sh, sc), rows in agg.items():
    m = [sum(r[j] for r in rows) / len(rows) for j in range(5)]
    print(f"{sh:20s} sched {sc:2d}: dur {m[0]:8.1f} us  " + "  ".join(f"{n}={x:.5g}" for n, x in zip(names, m[1:])) +
          f"  MFMA_BUSY/WAVE_CYCLES={m[2] / m[3]:.4f}  GUI/dur={m[4] / m[0] / 1e3:.3f} (x 8 XCDs) GHz")
