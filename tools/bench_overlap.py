"""Do two independent fit GEMMs overlap on two streams? (wgrad2 || dgrad2, h-branch || field branch)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
from dvt_amd import _lib  # noqa: E402

L = _lib.lib()
dev = "cuda"
B = 2048


def mk(n, k):
    return dict(x=torch.randn(B, k, device=dev), w=torch.randn(n, k, device=dev), b=torch.randn(n, device=dev),
                y=torch.empty(B, n, device=dev), dy=torch.randn(B, n, device=dev), dw=torch.zeros(n, k, device=dev),
                db=torch.zeros(n, device=dev), dx=torch.empty(B, k, device=dev), n=n, k=k)


f2, h1, h3 = mk(768, 384), mk(192, 768), mk(768, 192)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def wgrad(t, st):
    L.dvt_linear_bwd(t["dy"].data_ptr(), t["x"].data_ptr(), t["w"].data_ptr(), t["dw"].data_ptr(), t["db"].data_ptr(),
                     None, None, B, t["n"], t["k"], st.cuda_stream)


def dgrad(t, st):
    L.dvt_linear_bwd(t["dy"].data_ptr(), None, t["w"].data_ptr(), None, None, t["dx"].data_ptr(), t["x"].data_ptr(),
                     B, t["n"], t["k"], st.cuda_stream)


def fwd(t, st):
    L.dvt_linear_fwd(t["x"].data_ptr(), t["w"].data_ptr(), t["b"].data_ptr(), t["y"].data_ptr(), B, t["n"], t["k"], 1,
                     st.cuda_stream)


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s1)
    for _ in range(n):
        fn()
    e = torch.cuda.Event()
    e.record(s2)
    s1.wait_event(e)
    b.record(s1)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def pair(fa, fb):
    def seq():
        fa(s1); fb(s1)
    def par():
        e = torch.cuda.Event(); e.record(s1); s2.wait_event(e)
        fa(s1); fb(s2)
        e2 = torch.cuda.Event(); e2.record(s2); s1.wait_event(e2)
    return timeit(seq), timeit(par)


print("wgrad2 || dgrad2      seq %.1f us  par %.1f us" % pair(lambda s: wgrad(f2, s), lambda s: dgrad(f2, s)))
print("fwd field2 || fwd h1  seq %.1f us  par %.1f us" % pair(lambda s: fwd(f2, s), lambda s: fwd(h1, s)))
print("dgrad2 || (wgrad_h3+dgrad_h3) seq %.1f us par %.1f us" % pair(lambda s: dgrad(f2, s), lambda s: (wgrad(h3, s), dgrad(h3, s))))
