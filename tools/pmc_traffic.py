"""Reduce two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE; summaries written by tools/pmc_stats.py, see
`bash tools/gpu.sh pmc`) to the per-probe HBM-side bytes per launch that bench.py reports as `traffic`.

    python tools/pmc_traffic.py <dir with FETCH_SIZE.txt and WRITE_SIZE.txt> profiles/r03/pmc_traffic.json

bytes = (FETCH_SIZE x 2 + WRITE_SIZE) x 1024 / dispatches.  FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950
FETCH_SIZE counts a wide coalesced read at half its bytes (MI355X_MICROARCH.md, HBM section) -- hence x 2.  The
calibration row is the final layernorm kernel, whose algorithmic traffic is known exactly (reads fp32 rows, writes fp32
rows of the kept tokens).  FETCH_SIZE counts L2 misses (memory-side-cache hits included), so `traffic` is an upper
bound of the HBM bytes.
"""
import json
import re
import sys

PROBE_OF = [  # first match wins
    (r"gemm_bf16_kernel", "vit_gemm"), (r"attention_kernel", "vit_attn"), (r"adam_dense", "adam"),
    (r"fit_rows_kernel", "fit_rows"), (r"fit_backward_kernel", "fit_gemm"), (r"grid_sort_kernel", "grid"),
]


def parse(path):
    out, name, n = {}, None, 0
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+dispatches=(\d+)", line)
        if m:
            name, n = m.group(1), int(m.group(2))
            continue
        m = re.match(r"^\s+(FETCH_SIZE|WRITE_SIZE)\s+total=([0-9.e+]+)", line)
        if m and name:
            out[name] = (float(m.group(2)), n)
    return out


d, dst = sys.argv[1], sys.argv[2]
fetch, write = parse(f"{d}/FETCH_SIZE.txt"), parse(f"{d}/WRITE_SIZE.txt")
probes, kernels = {}, {}
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, (0.0, 0))
    w, nw = write.get(k, (0.0, 0))
    n = max(nf, nw)
    if n == 0:
        continue
    kernels[k[:100]] = {"dispatches": n, "fetch_kib_per_launch": f / n, "write_kib_per_launch": w / n,
                        "bytes_per_launch": (2.0 * f + w) * 1024.0 / n}
    for pat, probe in PROBE_OF:
        if re.search(pat, k):
            p = probes.setdefault(probe, {"bytes": 0.0, "launches": 0})
            p["bytes"] += (2.0 * f + w) * 1024.0
            p["launches"] += n
            break
views = sys.argv[3] if len(sys.argv) > 3 else "385"
doc = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over tools/pmc_target.py (ONE extractor launch of "
                 f"{views} views -- the bench runs 398 + 371 per image -- + 60 fit steps; DVT_PMC_ARGS={views} bash tools/gpu.sh pmc)",
       "formula": "(FETCH_SIZE x 2 + WRITE_SIZE) x 1024 B / dispatches",
       "extract_launch_views": int(views),  # bench.py scales the ViT kernels' bytes to the views of ITS launches
       "probes": {k: {"bytes_per_launch": v["bytes"] / v["launches"], "launches": v["launches"]} for k, v in probes.items()},
       "kernels": kernels}
open(dst, "w").write(json.dumps(doc, indent=1))
for k, v in doc["probes"].items():
    print(f"{k:10s} {v['bytes_per_launch'] / 1e6:9.1f} MB per launch over {v['launches']} launches")
