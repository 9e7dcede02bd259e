"""GPU: the stage-1 driver end to end (reference main_img_denoising.py flow) on a real image
file: CLI flags, view synthesis + coordinates, extractor, pipelined fit, output layout
(`raw_features/<model>/...npy` [H,W,C], `denoised_features/...npy` [1,H,W,C]), resume."""
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make_image(path, h=300, w=400):
    from PIL import Image
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 255 / w), (yy * 255 / h), ((xx // 20 + yy // 20) % 2) * 200], -1).astype(np.uint8)
    Image.fromarray(img).save(path)


@pytest.mark.parametrize("fit_batch", [1, 2])
def test_stage1_driver_end_to_end(built_lib, tmp_path, capsys, fit_batch):
    """three images: with fit_batch=2 one full group (shared launches) and one partial group"""
    from dvt_amd import stage1
    data_root = tmp_path / "data"
    (data_root / "sub").mkdir(parents=True)
    for name in ("sub/a.png", "b.png", "c.png"):
        _make_image(str(data_root / name))
    lst = tmp_path / "list.txt"
    lst.write_text("sub/a.png\nb.png extra-token\nc.png\n")
    argv = ["--img_path", str(lst), "--data_root", str(data_root), "--save_root", str(tmp_path / "out"),
            "--output_dir", str(tmp_path / "work"), "--num_views", "63", "--num_iters", "60",
            "--warmup_iters", "6", "--pixel_bsz", "512", "--num_imgs", "10", "--fit_batch", str(fit_batch),
            "--allow_random_vit"]
    args = stage1.get_args(argv)
    assert args.input_size == (518, 518) and args.n_levels == 16 and args.model.startswith("vit_base")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # random ViT weights (no checkpoint offline)
        done = stage1.main(args)
    assert done == 3
    model = "vit_base_patch14_dinov2.lvd142m"
    for rel in ("sub/a.npy", "b.npy", "c.npy"):
        raw = np.load(tmp_path / "out" / "raw_features" / model / rel)
        den = np.load(tmp_path / "out" / "denoised_features" / model / rel)
        assert raw.shape == (37, 37, 768) and raw.dtype == np.float32
        assert den.shape == (1, 37, 37, 768) and den.dtype == np.float32  # stage-2 loader squeezes it
        assert np.isfinite(raw).all() and np.isfinite(den).all()
        assert np.abs(den).max() > 0
    # resume: both outputs exist -> skipped (misc.check_if_file_exists)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert stage1.main(args) == 0
    assert "Skipping" in capsys.readouterr().out


def test_stage1_driver_fp32_matmul_high_equals_highest(built_lib, tmp_path):
    """`--dtype float32 --fp32_matmul high` (bf16x3 linear layers in the extractor, an opt-in) against the default
    `--fp32_matmul highest` on the same image, same seeds: raw features to bf16x3 accuracy, denoised features inside the
    fp32 chain bar; and the switch is ignored (not an error) under --dtype bfloat16."""
    from dvt_amd import stage1
    data_root = tmp_path / "data"
    data_root.mkdir()
    _make_image(str(data_root / "a.png"))
    lst = tmp_path / "list.txt"
    lst.write_text("a.png\n")
    outs = {}
    for mm in ("highest", "high"):
        argv = ["--img_path", str(lst), "--data_root", str(data_root), "--save_root", str(tmp_path / mm),
                "--output_dir", str(tmp_path / ("work_" + mm)), "--num_views", "15", "--num_iters", "60", "--warmup_iters", "6",
                "--num_imgs", "1", "--allow_random_vit", "--dtype", "float32", "--fp32_matmul", mm]
        args = stage1.get_args(argv)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert stage1.main(args) == 1
        model = "vit_base_patch14_dinov2.lvd142m"
        outs[mm] = (np.load(tmp_path / mm / "raw_features" / model / "a.npy"),
                    np.load(tmp_path / mm / "denoised_features" / model / "a.npy")[0])
    raw_a, raw_b = (torch.from_numpy(outs[m][0]).reshape(-1, 768) for m in ("highest", "high"))
    den_a, den_b = (torch.from_numpy(outs[m][1]).reshape(-1, 768) for m in ("highest", "high"))
    raw_err = float((raw_a - raw_b).norm() / raw_a.norm())
    cos = torch.nn.functional.cosine_similarity(den_a, den_b, dim=-1)
    print(f"[--fp32_matmul high vs highest] raw features rel-L2 {raw_err:.2e}; denoised cos mean {cos.mean():.6f} min {cos.min():.6f}")
    assert 0 < raw_err < 1e-4
    assert cos.mean() >= 0.9999 and cos.min() >= 0.999
    args = stage1.get_args(["--img_path", str(lst), "--data_root", str(data_root), "--save_root", str(tmp_path / "bf"),
                            "--num_views", "15", "--num_iters", "20", "--num_imgs", "1", "--allow_random_vit",
                            "--dtype", "bfloat16", "--fp32_matmul", "high"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert stage1.main(args) == 1


def test_render_views_vs_torch_antialias_bicubic(built_lib):
    """dvt_render_views against torch.nn.functional.interpolate(mode="bicubic", antialias=True) (the
    op SURVEY.md 8c names as the oracle for transform.py:50-52): up-sampled random crops, border
    crops, flips, the identity box and a down-scaling base resize."""
    from dvt_amd import views as V
    import torch.nn.functional as F
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    img = torch.randn(3, 518, 518, generator=g)
    rng = np.random.RandomState(1)
    boxes, _ = V.sample_view_boxes(40, (518, 518), 37, 37, rng)
    boxes = np.concatenate([boxes, [[0, 0, 200, 260, 1], [318, 258, 200, 260, 0], [100, 0, 418, 518, 1]]])
    out = torch.empty(len(boxes), 3, 518, 518, device=dev)
    V.render_views(img.to(dev), boxes, out)
    got = out.cpu()
    for v, (i, j, h, w, flip) in enumerate(boxes.tolist()):
        ref = F.interpolate(img[None, :, i:i + h, j:j + w], size=(518, 518), mode="bicubic",
                            antialias=True, align_corners=False)[0]
        if flip:
            ref = ref.flip(-1)
        err = float((got[v] - ref).abs().max())
        assert err < 2e-5, (v, (i, j, h, w, flip), err)
    assert float((got[40] - img).abs().max()) < 1e-6  # the full-image box is the identity
    # base resize with down-scaling (single_image_dataset.py:33-38): 900x700 -> 518x518
    big = torch.randn(3, 900, 700, generator=g)
    o2 = torch.empty(1, 3, 518, 518, device=dev)
    V.render_views(big.to(dev), np.array([[0, 0, 900, 700, 0]]), o2)
    ref = F.interpolate(big[None], size=(518, 518), mode="bicubic", antialias=True, align_corners=False)[0]
    assert float((o2[0].cpu() - ref).abs().max()) < 2e-5


def test_vit_large_geometry(built_lib):
    """BASELINE configs[2]: DINOv2 ViT-L/14 (dim 1024, 16 heads) through the same kernels;
    2 blocks of random well-conditioned weights against the fp32 oracle."""
    from dvt_amd.vit import HipViT, random_state_dict
    from oracle import vit as ovit
    sd = random_state_dict(1024, 2, 14, 1370, seed=7, well_conditioned=True)
    x = torch.randn(2, 3, 518, 518, generator=torch.Generator().manual_seed(2))
    want = ovit.forward_features(sd, x, 14, 14)
    got = HipViT(sd, 14, 14, (518, 518), "cuda").forward_features(x.cuda()).cpu()
    cos = torch.nn.functional.cosine_similarity(got.reshape(-1, 1024), want.reshape(-1, 1024), dim=-1)
    print(f"ViT-L/14 geometry: cos mean {cos.mean():.6f} min {cos.min():.6f}")
    assert got.shape == (2, 37, 37, 1024) and cos.min() > 0.999


def test_fit_large_feature_dim(built_lib):
    """ViT-L feature width (C = 1024: MLP 128->512->1024, h 1024->256->256->1024) through the fused loop against the
    ORACLE loop (oracle/fit.py == reference main_img_denoising.py:28-149), same initial parameters and index stream,
    both precisions, across the phase switch: per-step losses and the saved tensor.  (The 1000-step schedule at this
    width is tests/test_gpu_parity_full.py::test_fit_baseline_schedule_vs_oracle_fixture[1024].)"""
    from oracle import fit as ofit
    from tests.test_gpu_fit import per_patch_cos, synthetic_image
    from tests.test_gpu_parity_full import hip_engine_from, oracle_modules
    V_, H, W, C, T, WARM = 6, 37, 37, 1024, 40, 4
    feats, xy = synthetic_image(V_, H, W, C, seed=4)
    n_rows = V_ * H * W
    idx = np.random.RandomState(4).randint(0, n_rows, (T, 2048)).astype(np.int32)
    d_o, f_o = oracle_modules(3, H, W, C)
    res = {}
    for mode in ("float32", "bfloat16"):
        eng = hip_engine_from(d_o, f_o, n_rows, T, WARM, mode, H=H, W=W, C=C)
        eng.fit(feats.reshape(-1, C).cuda(), xy.reshape(-1, 2).cuda(), idx, log_every=1)
        torch.cuda.synchronize()
        res[mode] = (eng.loss_log(), eng.infer(xy[-1].cuda()).cpu())
        assert float(eng.grads.abs().max()) == 0.0
        del eng
    want_log = ofit.fit_image(d_o, f_o, feats, xy, idx, num_iters=T, warmup_iters=WARM, log_every=1)
    want = ofit.final_denoised_feats(d_o, f_o, feats, xy)[0]
    for mode, tol in (("float32", 1e-3), ("bfloat16", 3e-2)):
        log, got = res[mode]
        assert got.shape == (37, 37, 1024) and len(log) == T
        worst = max(abs(log[s][k] - v) / max(1.0, abs(v)) for s in range(T) for k, v in want_log[s].items())
        cos = per_patch_cos(got, want)
        print(f"[C=1024, {mode} fit, {T} steps] worst per-step loss rel err {worst:.2e}; denoised_feats cosine mean "
              f"{cos.mean():.6f} min {cos.min():.6f}")
        assert worst <= tol, (mode, worst)
        assert cos.mean() >= 0.999 and cos.min() >= 0.99, (mode, float(cos.mean()), float(cos.min()))
    assert want_log[T - 1]["patch_l2_loss"] < want_log[0]["patch_l2_loss"]


def test_pipeline_consumes_the_numpy_stream_in_image_order(built_lib, monkeypatch):
    """The pipelined driver draws index streams on a look-ahead thread; every image must still get the
    draws the reference's sequential loop would give it (np.random.randint after fix_random_seeds,
    main_img_denoising.py:73), across two consecutive run() calls, and the pipelined results must equal
    the strictly serial flow (depth 1) up to fp32 atomics order."""
    from types import SimpleNamespace
    from dvt_amd import fit as fit_mod
    from dvt_amd import stage1
    from dvt_amd import views as V
    from dvt_amd.utils import misc
    args = SimpleNamespace(model="vit_base_patch14_dinov2.lvd142m", input_size=(518, 518), stride_size=14,
                           layer_depth_ratio=1.0, num_views=7, num_iters=24, warmup_iters=2, n_levels=16,
                           freeze_shared_artifacts_after=0.5, lr=0.01, min_lr=0.001, weight_decay=1e-5,
                           extract_bsz=32, pixel_bsz=256, seed=0, vit_checkpoint=None, dtype="float32",
                           allow_random_vit=True)
    n_img = 5

    def run(depth):
        misc.fix_random_seeds(0)
        seen = []
        real = fit_mod.FitEngine.buffers

        def spy(self, feat, xy, idx=None, log_every=1000):
            seen.append(None if idx is None else np.array(idx, copy=True))
            return real(self, feat, xy, idx, log_every)

        monkeypatch.setattr(fit_mod.FitEngine, "buffers", spy)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            st = stage1.Stage1(args, torch.device("cuda"), depth=depth)
        outs = []

        def jobs(lo, hi):
            for k in range(lo, hi):
                def set_views(slot, k=k):
                    v, c = V.synthetic_views(args.num_views, args.input_size, st.pos_h, st.pos_w,
                                             torch.device("cuda"), seed=k)
                    slot.views.copy_(v)
                    slot.coords.copy_(c)
                yield k, set_views

        def on_result(tag, raw, den):
            outs.append((tag, den.copy()))

        assert st.run(jobs(0, 3), on_result) == 3
        assert st.run(jobs(3, n_img), on_result) == n_img - 3
        monkeypatch.setattr(fit_mod.FitEngine, "buffers", real)
        return seen, dict(outs)

    seen, pipe = run(depth=2)
    misc.fix_random_seeds(0)
    n_rows = (args.num_views + 1) * 37 * 37
    for k in range(n_img):
        want = np.random.randint(0, n_rows, (args.num_iters, args.pixel_bsz)).astype(np.int32)
        assert seen[k] is not None and np.array_equal(seen[k], want), k
    _, serial = run(depth=1)
    for k in range(n_img):
        a, b = torch.from_numpy(pipe[k]).reshape(-1, 768), torch.from_numpy(serial[k]).reshape(-1, 768)
        cos = torch.nn.functional.cosine_similarity(a, b, dim=-1)
        assert cos.mean() > 0.9999 and cos.min() > 0.999, (k, float(cos.mean()), float(cos.min()))


def test_stage1_driver_stride7_register_backbone(built_lib, tmp_path):
    """SURVEY N4 end to end: the reference demo's `--stride_size 7` (73 x 73 lattice, 5 329 + 5 tokens per
    view, pos_embed resampled from 37 x 37) with a register-token backbone through the whole driver."""
    from dvt_amd import stage1
    data_root = tmp_path / "data"
    data_root.mkdir()
    _make_image(str(data_root / "a.png"))
    lst = tmp_path / "list.txt"
    lst.write_text("a.png\n")
    model = "vit_small_patch14_reg4_dinov2.lvd142m"
    argv = ["--img_path", str(lst), "--data_root", str(data_root), "--save_root", str(tmp_path / "out"),
            "--output_dir", str(tmp_path / "work"), "--model", model, "--stride_size", "7", "--num_views", "7",
            "--num_iters", "40", "--warmup_iters", "4", "--pixel_bsz", "512", "--num_imgs", "1", "--allow_random_vit"]
    args = stage1.get_args(argv)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert stage1.main(args) == 1
    raw = np.load(tmp_path / "out" / "raw_features" / model / "a.npy")
    den = np.load(tmp_path / "out" / "denoised_features" / model / "a.npy")
    assert raw.shape == (73, 73, 384) and den.shape == (1, 73, 73, 384)
    assert np.isfinite(raw).all() and np.isfinite(den).all() and np.abs(den).max() > 0
