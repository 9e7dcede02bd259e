"""Build the DEVELOPER library denoising-vit_amd/csrc/libdvt_hip_lab.so (-DDVT_LAB): the product sources plus csrc/lab/ --
superseded GEMM schedules (256x128 lock-step, 256x256 two-stage), the 4-wave persistent GEMM with its ablations, the 8p
re-schedules and timing builds, the round-2 attention loop and the attention schedule masks.  Loaded only by tools/lab_*.py
and tests/test_gpu_lab.py; `dvt_amd` and bench.py always load the product library.

    python tools/build_lab.py [--force]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "denoising-vit_amd"))
from dvt_amd import _lib  # noqa: E402

if __name__ == "__main__":
    t0 = time.time()
    path = _lib.build(force="--force" in sys.argv, verbose=True, lab=True)
    print(f"{path}: {os.path.getsize(path) / 1e6:.2f} MB in {time.time() - t0:.0f} s")
