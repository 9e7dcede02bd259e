"""Developer tool (BASELINE configs[2]): fit throughput vs the number of concurrent neural fields K on one GPU.
K fits (independent arenas / Adam state / index streams; one shared synthetic feature store) advance as groups of
<= FIT_BATCH_MAX with shared launches (dvt_fit_run_batched), the groups on separate HIP streams fed by one host
thread each.  Prints us per step per fit.

    python tools/bench_fit_batch.py [C=768] [steps=300]
"""
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd.fit import FIT_BATCH_MAX, FitEngine, FitSettings, fit_many  # noqa: E402

C_ = int(sys.argv[1]) if len(sys.argv) > 1 else 768
T = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
n_rows = 769 * 1369
g = torch.Generator(device=dev).manual_seed(0)
feat = torch.randn(n_rows, C_, device=dev, generator=g)
xy = torch.rand(n_rows, 2, device=dev, generator=g)
s = FitSettings(feat_dim=C_, num_iters=T, warmup_iters=T // 10, mlp_dtype="bfloat16")
KMAX = 16
from dvt_amd import _lib  # noqa: E402
for kv in filter(None, os.environ.get("DVT_TUNE", "").split(",")):  # e.g. DVT_TUNE=13=0
    k_, v_ = kv.split("=")
    assert _lib.lib().dvt_tune_set(int(k_), int(v_)) == 0
engines = [FitEngine(s, n_rows, dev) for _ in range(KMAX)]
idxs = [np.random.RandomState(j).randint(0, n_rows, (T, 2048)).astype(np.int32) for j in range(KMAX)]
streams = [torch.cuda.Stream(device=dev) for _ in range(KMAX)]


def run(K, per_group):
    groups = [list(range(i, min(K, i + per_group))) for i in range(0, K, per_group)]
    for e in engines[:K]:
        e.reset(g)
    torch.cuda.synchronize()

    def work(gi, members):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[gi]):
            fit_many([engines[j] for j in members], [feat] * len(members), [xy] * len(members),
                     [idxs[j] for j in members], log_every=0)

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(gi, m)) for gi, m in enumerate(groups)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / T / K * 1e6


CASES = [(1, 1), (2, 2), (4, 4), (2, 1), (4, 1), (4, 2), (8, 4), (8, 2), (8, 1), (16, 4)]
if os.environ.get("DVT_FB_CASES"):  # e.g. "4:4,1:1" (profiling runs)
    CASES = [tuple(int(v) for v in c.split(":")) for c in os.environ["DVT_FB_CASES"].split(",")]
for K, per in CASES:
    if per > FIT_BATCH_MAX:
        continue
    run(K, per)
    us = min(run(K, per) for _ in range(2))
    print(f"C={C_} K={K:2d} fits as {-(-K // per)} stream(s) x {per} fits per launch: {us:7.1f} us per step per fit "
          f"({us * K:7.1f} us per K-step)", flush=True)
