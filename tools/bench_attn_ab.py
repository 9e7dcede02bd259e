"""Interleaved A/B of the bf16 extractor's attention entry points at the bench's launch shape (round 6).

    python tools/bench_attn_ab.py [--views 396] [--s_pad 1376] [--rounds 6] [--lab]

raw   = dvt_vit_attention       (q as it is, attention_kernel_v2<15>: rounds 3-5)
log2q = dvt_vit_attention_log2q (q pre-scaled by log2(e) / 8, attention_kernel_l2<559>: round 6)
--lab adds the ablation builds of the developer library (idle waves not skipped / no half tail tile / neither).
Variants alternate inside one process, the first round is dropped (the clock ramps over the first launches,
profiles/r06/README.md); times are hipEvent brackets around `reps` back-to-back launches on one stream.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "denoising-vit_amd"))

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=396)
ap.add_argument("--heads", type=int, default=12)
ap.add_argument("--s_pad", type=int, default=1376)
ap.add_argument("--n_valid", type=int, default=1370)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--lab", action="store_true")
ap.add_argument("--only", default="", help="with --lab: comma-separated toggles x of dvt_tune_set(1, -540 - x) to compare (0 = the product's mask)")
a = ap.parse_args()

if a.lab:
    from tools import labenv
    L = labenv.use_lab_library()
else:
    from dvt_amd import _lib
    import dvt_amd.vit  # noqa: F401
    L = _lib.lib()

dev = "cuda"
torch.manual_seed(0)
dim = a.heads * 64
rows = a.views * a.s_pad
qk = torch.randn(rows + 128, 2 * dim, device=dev).bfloat16()
qk_l2 = qk.clone()
qk_l2[:, :dim] = (qk[:, :dim].float() * (0.125 * 1.4426950408889634)).bfloat16()
vt = torch.randn(a.views, a.heads, 64, a.s_pad, device=dev).bfloat16()
out = torch.empty(rows + 128, dim, device=dev, dtype=torch.bfloat16)
S = torch.cuda.current_stream().cuda_stream
flops = 4.0 * a.n_valid * a.n_valid * 64 * a.heads * a.views


def raw():
    assert L.dvt_vit_attention(qk.data_ptr(), vt.data_ptr(), out.data_ptr(), a.views, a.heads, a.s_pad, a.n_valid, S) == 0


def l2(extra=0):
    def f():
        if a.lab:
            assert L.dvt_tune_set(1, -540 - extra) == 0
        assert L.dvt_vit_attention_log2q(qk_l2.data_ptr(), vt.data_ptr(), out.data_ptr(), a.views, a.heads, a.s_pad, a.n_valid, S) == 0
    return f


variants = [("raw  v2<15>", raw), ("log2q l2<559>", l2(0))]
if a.lab:
    variants += [("log2q, P.V fragment by fragment (557)", l2(2)), ("log2q, K reads unplaced (47)", l2(512)), ("log2q, both (45)", l2(514)),
                 ("log2q, idle waves compute (559+128)", l2(128)), ("log2q, whole tail tile (559+256)", l2(256)),
                 ("log2q, neither (559+384)", l2(384)), ("log2q, P packed behind the V^T reads (559+1024)", l2(1024)),
                 ("log2q, s_setprio 1 around P.V (559+2048)", l2(2048)), ("log2q, s_setprio 1 around S + softmax (559+4096)", l2(4096)),
                 ("TIMING ONLY: log2q without the tile barrier (559+8192)", l2(8192)),
                 ("log2q, softmax split over the two blocks (559+16384)", l2(16384)),
                 ("log2q, iglp_opt(2) MFMAExpInterleave for the softmax block", l2(32768)),
                 ("log2q, iglp_opt(3) for the softmax block", l2(32768 + 65536))]
if a.lab and a.only:
    variants = [(f"log2q, mask 559 ^ {int(x)}", l2(int(x))) for x in a.only.split(",")]
times = {n: [] for n, _ in variants}
for r in range(a.rounds + 1):
    for name, fn in variants:
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if r > 0:
            times[name].append(e0.elapsed_time(e1) * 1e3 / a.reps)
if a.lab:
    L.dvt_tune_set(1, -540)
print(f"attention, {a.views} views x {a.heads} heads, s_pad {a.s_pad}, {a.n_valid} tokens; us per launch (median of {a.rounds} interleaved rounds), TF/s")
base = None
for name, _ in variants:
    t = sorted(times[name])[len(times[name]) // 2]
    base = base or t
    print(f"  {name:38s} {t:9.1f} us  {flops / t / 1e6:7.1f} TF/s  {100 * (t / base - 1):+6.2f} %   [{min(times[name]):.1f} .. {max(times[name]):.1f}]")
