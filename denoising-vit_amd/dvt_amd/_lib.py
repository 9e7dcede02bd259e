"""ctypes binding of libdvt_hip.so (the C ABI declared in include/dvt_hip.h).

This is the ONLY place the Python host side touches native code.  There is deliberately no
CPU fallback: every compute entry point needs CUDA(HIP) tensors and raises if the library
or a GPU is missing -- the product path never routes through `oracle/`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import torch

CSRC = Path(__file__).resolve().parent.parent / "csrc"
LIB_PATH = CSRC / "libdvt_hip.so"
LAB_LIB_PATH = CSRC / "libdvt_hip_lab.so"  # developer build (-DDVT_LAB), see build(lab=True); never loaded by dvt_amd
HIP_SOURCES = [
    "dvt_grid.hip",
    "dvt_gemm_f32.hip",
    "dvt_loss.hip",
    "dvt_adam.hip",
    "dvt_fit.hip",
    "dvt_fit_fused.hip",
    "dvt_vit.hip",
    "dvt_vit_f32.hip",
    "dvt_stage2.hip",
    "dvt_prof.hip",
    "dvt_views.hip",
]
DVT_MAX_LEVELS = 32
DVT_ADAM_MAX_SEGS = 8


class DvtError(RuntimeError):
    pass


# --------------------------------------------------------------------------- struct mirrors
class GridTable(C.Structure):
    _fields_ = [
        ("n_levels", C.c_int32),
        ("n_features", C.c_int32),
        ("n_entries_total", C.c_uint32),
        ("pad_", C.c_uint32),
        ("scale", C.c_float * DVT_MAX_LEVELS),
        ("resolution", C.c_uint32 * DVT_MAX_LEVELS),
        ("entries", C.c_uint32 * DVT_MAX_LEVELS),
        ("offset", C.c_uint32 * DVT_MAX_LEVELS),
        ("hashed", C.c_uint32 * DVT_MAX_LEVELS),
    ]


class AdamSeg(C.Structure):
    _fields_ = [
        ("begin", C.c_int64),
        ("end", C.c_int64),
        ("lr", C.c_double),
        ("bias_correction1", C.c_double),
        ("bias_correction2_sqrt", C.c_double),
        ("active", C.c_int32),
        ("pad_", C.c_int32),
    ]


class AdamArgs(C.Structure):
    _fields_ = [
        ("beta1", C.c_double),
        ("beta2", C.c_double),
        ("eps", C.c_double),
        ("weight_decay", C.c_double),
        ("n_segs", C.c_int32),
        ("pad_", C.c_int32),
        ("sparse_end", C.c_int64),
        ("segs", AdamSeg * DVT_ADAM_MAX_SEGS),
    ]


class FitConfig(C.Structure):
    _fields_ = [
        ("feat_dim", C.c_int32),
        ("hidden", C.c_int32),
        ("res_hidden", C.c_int32),
        ("lattice", C.c_int32),
        ("n_rows", C.c_int32),
        ("batch", C.c_int32),
        ("num_iters", C.c_int32),
        ("switch_step", C.c_int32),
        ("enable_residual", C.c_int32),
        ("mlp_bf16", C.c_int32),
        ("grad_scale", C.c_double),
        ("beta1", C.c_double),
        ("beta2", C.c_double),
        ("eps", C.c_double),
        ("weight_decay", C.c_double),
        ("grid", GridTable),
        ("off_grid", C.c_int64),
        ("off_w1", C.c_int64),
        ("off_b1", C.c_int64),
        ("off_w2", C.c_int64),
        ("off_b2", C.c_int64),
        ("off_G", C.c_int64),
        ("off_wh1", C.c_int64),
        ("off_bh1", C.c_int64),
        ("off_wh2", C.c_int64),
        ("off_bh2", C.c_int64),
        ("off_wh3", C.c_int64),
        ("off_bh3", C.c_int64),
        ("arena_floats", C.c_int64),
    ]


class FitBuffers(C.Structure):
    _fields_ = [
        ("feat", C.c_void_p),
        ("xy", C.c_void_p),
        ("idx", C.c_void_p),
        ("params", C.c_void_p),
        ("adam_m", C.c_void_p),
        ("adam_v", C.c_void_p),
        ("grads", C.c_void_p),
        ("touched", C.c_void_p),
        ("workspace", C.c_void_p),
        ("losses", C.c_void_p),
        ("h_lr", C.c_void_p),
        ("log_every", C.c_int32),
        ("pad_", C.c_int32),
    ]


_P = C.c_void_p
_I = C.c_int
_SIGNATURES = {
    # name: (restype, argtypes)
    "dvt_abi_version": (_I, []),
    "dvt_struct_sizes": (_I, [C.POINTER(C.c_int64)]),
    "dvt_grid_table": (_I, [_I, _I, _I, _I, _I, C.POINTER(GridTable)]),
    "dvt_grid_fwd": (_I, [C.POINTER(GridTable), _P, _P, _P, _I, _P]),
    "dvt_grid_bwd": (_I, [C.POINTER(GridTable), _P, _P, _P, _P, _I, _P]),
    "dvt_grid_corners": (_I, [C.POINTER(GridTable), _P, _P, _P, _I, _P]),
    "dvt_linear_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dvt_linear_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "dvt_gather_rows": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "dvt_scatter_add_rows": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "dvt_bilinear_rows_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dvt_bilinear_rows_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dvt_loss_fwd_bwd": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, C.c_float, _P]),
    "dvt_loss_reduce": (_I, [_P, _P, _I, _I, _I, _P]),
    "dvt_adam_step": (_I, [C.POINTER(AdamArgs), _P, _P, _P, _P, _P, _P]),
    "dvt_fit_layout": (_I, [C.POINTER(FitConfig)]),
    "dvt_fit_workspace_floats": (C.c_int64, [C.POINTER(FitConfig)]),
    "dvt_fit_run": (_I, [C.POINTER(FitConfig), C.POINTER(FitBuffers), _I, _I, _P]),
    "dvt_fit_run_batched": (_I, [C.POINTER(FitConfig), _I, C.POINTER(C.POINTER(FitBuffers)), _I, _I, _P]),
    "dvt_field_infer": (_I, [C.POINTER(FitConfig), _P, _P, _P, _P, _I, _P]),
    "dvt_vit_gemm_residual": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "dvt_render_views": (_I, [_P, _I, _I, _P, _P, _I, _I, _I, _P]),
    "dvt_tune_set": (_I, [_I, _I]),
    "dvt_prof_enable": (_I, [C.c_uint]),
    "dvt_prof_collect": (_I, [_I, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
}

PROBES = {"adam": 0, "vit_gemm": 1, "vit_attn": 2, "fit_gemm": 3, "grid": 4, "fit_rows": 5}


user_tune: dict[int, int] = {}  # dvt_tune_set keys a USER set through tune() (bench.py --tune, developer tools): drivers leave them alone


def tune(key: int, value: int) -> None:
    """dvt_tune_set on behalf of the user: recorded, so that code which flips a knob per mode (Stage1.fit_group: key 6) does
    not silently override an explicit A/B setting."""
    check(lib().dvt_tune_set(int(key), int(value)), f"dvt_tune_set({key},{value})")
    user_tune[int(key)] = int(value)


def prof_enable(names=()) -> None:
    mask = 0
    for n in names:
        mask |= 1 << PROBES[n]
    check(lib().dvt_prof_enable(mask), "dvt_prof_enable")


def prof_collect(name: str) -> dict:
    ms, cnt, work = C.c_double(), C.c_int64(), C.c_double()
    check(lib().dvt_prof_collect(PROBES[name], C.byref(ms), C.byref(cnt), C.byref(work)),
          "dvt_prof_collect")
    return {"total_ms": ms.value, "launches": cnt.value, "work": work.value}

_lib = None


def register_signatures(extra: dict) -> None:
    """Other binding modules (ViT) add their entry points here before the first load."""
    _SIGNATURES.update(extra)
    if _lib is not None:  # already loaded: type the new entry points right away
        for name, (res, args) in extra.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args


def build(force: bool = False, verbose: bool = False, lab: bool = False) -> Path:
    """Compile every HIP source for gfx950 into csrc/libdvt_hip.so (hipcc cross-compiles without a GPU): one object per
    source, compiled in parallel, then one link.  Rebuilds only when a source/header is newer than the library.
    lab=True builds the DEVELOPER library csrc/libdvt_hip_lab.so with -DDVT_LAB (superseded schedules, experiments and
    timing builds of csrc/lab/); nothing in dvt_amd loads it (tools/build_lab.py, tools/lab_*.py, tests/test_gpu_lab.py)."""
    from concurrent.futures import ThreadPoolExecutor
    out = LAB_LIB_PATH if lab else LIB_PATH
    srcs = [CSRC / s for s in HIP_SOURCES if (CSRC / s).exists()]
    deps = srcs + list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + list((CSRC.parent.parent / "include").glob("*.h"))
    if lab:
        deps += list((CSRC / "lab").glob("*.inc"))
    deps = [d for d in deps if d.exists()]
    if (not force and out.exists()
            and all(out.stat().st_mtime >= d.stat().st_mtime for d in deps)):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + (["-DDVT_LAB"] if lab else [])
    objdir = CSRC / "build" / ("lab" if lab else "product")
    objdir.mkdir(parents=True, exist_ok=True)

    def compile_one(src: Path) -> Path:
        obj = objdir / (src.stem + ".o")
        cmd = [hipcc, *flags, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=str(CSRC))
        return obj

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as pool:
        objs = list(pool.map(compile_one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[str(o) for o in objs], "-o", str(out)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=str(CSRC))
    return out


def open_library(path: Path) -> C.CDLL:
    """dlopen + type one build of the library (every declared symbol must exist; struct mirrors are verified)."""
    handle = C.CDLL(str(path))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    sizes = (C.c_int64 * 5)()
    handle.dvt_struct_sizes(sizes)
    mine = [C.sizeof(GridTable), C.sizeof(AdamSeg), C.sizeof(AdamArgs), C.sizeof(FitConfig),
            C.sizeof(FitBuffers)]
    if list(sizes) != mine:
        raise DvtError(f"struct layout mismatch between ctypes mirrors {mine} and C {list(sizes)}")
    return handle


def lib() -> C.CDLL:
    """Load (once) and type the PRODUCT shared library; fail loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise DvtError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    _lib = open_library(LIB_PATH)
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc == 0:
        return
    if rc < 0:
        raise DvtError(f"{what}: invalid argument (DVT_E {rc})")
    raise DvtError(f"{what}: HIP error {rc}")


def ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise DvtError(
                "dvt_amd kernels need HIP device tensors (got a CPU tensor); there is no CPU "
                "fallback on the product path -- use oracle/ only as a test reference")


def grid_table(n_levels: int, n_features: int, base_resolution: int, max_resolution: int,
               log2_hashmap_size: int) -> GridTable:
    t = GridTable()
    check(lib().dvt_grid_table(n_levels, n_features, base_resolution, max_resolution,
                               log2_hashmap_size, C.byref(t)), "dvt_grid_table")
    return t
