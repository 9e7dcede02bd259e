"""Developer tools that need csrc/lab/ kernels call use_lab_library() BEFORE the first `_lib.lib()`: it builds
csrc/libdvt_hip_lab.so (-DDVT_LAB) when stale and points THIS PROCESS's binding at it.  The package itself never does."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "denoising-vit_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def use_lab_library():
    from dvt_amd import _lib
    assert _lib._lib is None, "use_lab_library() must run before the product library is loaded"
    _lib.build(lab=True)
    _lib.LIB_PATH = _lib.LAB_LIB_PATH
    import dvt_amd.vit  # noqa: F401  (registers the ViT entry points)
    handle = _lib.lib()
    assert handle.dvt_vit_is_lab_build() == 1
    return handle
