"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol the headers
declare, its host-side functions agree with the oracle, and the product path refuses to run
without a GPU (no CPU fallback)."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest
import torch

from oracle import hashgrid as hg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"^\s*(?:int|int64_t)\s+(dvt_\w+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(built_lib):
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(built_lib, n), f"libdvt_hip.so lacks {n}"
    from dvt_amd import _lib
    assert set(_lib._SIGNATURES) == set(names), "ctypes signatures out of sync with include/*.h"
    assert built_lib.dvt_abi_version() == 2


def test_struct_mirrors_match(built_lib):
    from dvt_amd import _lib
    sizes = (C.c_int64 * 5)()
    assert built_lib.dvt_struct_sizes(sizes) == 0
    assert list(sizes) == [C.sizeof(_lib.GridTable), C.sizeof(_lib.AdamSeg), C.sizeof(_lib.AdamArgs),
                           C.sizeof(_lib.FitConfig), C.sizeof(_lib.FitBuffers)]


@pytest.mark.parametrize("cfg", [(16, 8, 16, 1024, 20), (10, 8, 16, 1024, 20), (4, 8, 16, 64, 12),
                                 (16, 8, 16, 1024, 14), (1, 8, 16, 16, 20)])
def test_host_grid_table_equals_oracle(built_lib, cfg):
    from dvt_amd import _lib
    t = _lib.grid_table(*cfg)
    o = hg.grid_table(*cfg)
    L = cfg[0]
    assert list(t.resolution)[:L] == o.resolution.tolist()
    assert list(t.entries)[:L] == o.entries.tolist()
    assert list(t.offset)[:L] == o.offset.tolist()
    assert [bool(x) for x in list(t.hashed)[:L]] == o.hashed.tolist()
    assert np.array_equal(np.asarray(list(t.scale)[:L], np.float32), o.scale)  # bit-exact fp32
    assert t.n_entries_total == o.n_entries_total


def test_bad_arguments_are_rejected(built_lib):
    from dvt_amd import _lib
    t = _lib.GridTable()
    assert built_lib.dvt_grid_table(0, 8, 16, 1024, 20, C.byref(t)) == -1
    assert built_lib.dvt_grid_table(16, 2, 16, 1024, 20, C.byref(t)) == -1  # only F=8 is built
    assert built_lib.dvt_grid_table(40, 8, 16, 1024, 20, C.byref(t)) == -1
    assert built_lib.dvt_linear_fwd(None, None, None, None, 4, 4, 4, 0, None) == -1
    assert built_lib.dvt_linear_fwd(1, 1, None, 1, 4, 6, 8, 0, None) == -1  # n % 4 != 0


def test_fit_layout_host(built_lib):
    from dvt_amd import _lib
    cfg = _lib.FitConfig()
    cfg.feat_dim, cfg.hidden, cfg.res_hidden = 768, 384, 192
    cfg.lattice, cfg.n_rows, cfg.batch, cfg.num_iters = 1369, 769 * 1369, 2048, 1000
    cfg.grid = _lib.grid_table(16, 8, 16, 1024, 20)
    assert built_lib.dvt_fit_layout(C.byref(cfg)) == 0
    offs = [cfg.off_grid, cfg.off_w1, cfg.off_b1, cfg.off_w2, cfg.off_b2, cfg.off_G, cfg.off_wh1,
            cfg.off_bh1, cfg.off_wh2, cfg.off_bh2, cfg.off_wh3, cfg.off_bh3, cfg.arena_floats]
    assert offs[0] == 0 and all(o % 256 == 0 for o in offs) and offs == sorted(offs)
    assert cfg.off_w1 == (19741760 + 255) // 256 * 256  # 2 467 720 entries * 8, padded to 256
    # 21 471 296 trainable floats (+ alignment padding only)
    assert 21471296 <= cfg.arena_floats < 21471296 + 12 * 256
    assert built_lib.dvt_fit_workspace_floats(C.byref(cfg)) > 2048 * 768 * 4


def test_no_cpu_fallback(built_lib):
    from dvt_amd import _lib
    from dvt_amd.fit import FitEngine, FitSettings
    from dvt_amd.models import NeuralFeatureField
    with pytest.raises(_lib.DvtError):
        FitEngine(FitSettings(num_iters=10, warmup_iters=1), n_rows=100, device="cpu")
    f = NeuralFeatureField(feat_dim=16, n_levels=4, max_resolution=64, log2_hashmap_size=10)
    assert f.neural_field.params.numel() == hg.grid_table(4, 8, 16, 64, 10).n_params
    assert list(f.state_dict().keys()) == ["neural_field.params", "mlp.0.weight", "mlp.0.bias",
                                           "mlp.2.weight", "mlp.2.bias"]
    with pytest.raises(_lib.DvtError):
        f(torch.rand(8, 2))  # CPU tensor: must fail loudly, never fall back


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "denoising-vit_amd")
    for path in glob.glob(os.path.join(pkg, "**", "*.py"), recursive=True):
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path


def test_shard_and_resume_host_logic(tmp_path):
    from argparse import Namespace
    from dvt_amd.utils import misc
    # 1-vs-8-way split covers the same work-list exactly once (stage1.sh:15-16)
    for n in (0, 5, 8, 100, 10000):
        for ws in (1, 2, 3, 8):
            spans = [misc.shard_range(7, n, r, ws) for r in range(ws)]
            assert spans[0][0] == 7 and spans[-1][1] == 7 + n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    args = Namespace(save_root=str(tmp_path / "out"), model="m", data_root=str(tmp_path / "data"))
    fn = str(tmp_path / "data" / "a" / "x.jpg")
    assert not misc.check_if_file_exists(args, fn)
    raw_p, den_p = misc.output_paths(args.save_root, args.model, args.data_root, fn)
    # (the reference's str.replace layout keeps a double slash when data_root has no trailing "/")
    assert os.path.normpath(raw_p).endswith("out/raw_features/m/a/x.npy")
    assert os.path.normpath(den_p).endswith("out/denoised_features/m/a/x.npy")
    misc.atomic_save_npy(raw_p, np.zeros((2, 2), np.float32))
    assert not misc.check_if_file_exists(args, fn)  # needs BOTH files
    misc.atomic_save_npy(den_p, np.ones((1, 2, 2), np.float32))
    assert misc.check_if_file_exists(args, fn)
    assert np.load(den_p).shape == (1, 2, 2)
