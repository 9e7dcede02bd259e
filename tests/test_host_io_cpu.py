"""CPU: host-side I/O contracts of the stage-1 path.

  * the base resize is the reference's (PIL bilinear on uint8, main_img_denoising.py:279-284), not a
    float bicubic one;
  * the product's view boxes / coordinates equal the oracle restatement of transform.py:39-76;
  * the .npy pair stage 1 writes (main_img_denoising.py:131-146) round-trips through the stage-2
    reader (dvt/dataset/paired_list_dataset.py:27-43) -- SURVEY.md 8c known-answer material;
  * the stage-1 CLI refuses to run on random ViT weights unless told so.
"""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import stage2_reader
from oracle import views as oviews


def _photo(h, w, seed=0):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([xx * 255.0 / w, yy * 255.0 / h, ((xx // 9 + yy // 7) % 2) * 180.0], -1)
    return np.clip(base + rng.randn(h, w, 3) * 20, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("h,w", [(375, 500), (1200, 900), (518, 518), (200, 640)])
def test_base_resize_is_pil_bilinear_on_uint8(h, w):
    from PIL import Image
    from dvt_amd import views as V
    img = _photo(h, w)
    got = V.base_resize_u8(img, (518, 518))
    want = np.asarray(Image.fromarray(img).resize((518, 518), Image.BILINEAR))
    assert got.dtype == np.uint8 and got.shape == (518, 518, 3) and np.array_equal(got, want)
    u8, x = oviews.base_transform(img, (518, 518), (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
    assert np.array_equal(u8, got)
    y = V.normalize_u8(got, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225), "cpu")
    assert torch.equal(x, y)
    if (h, w) != (518, 518):  # and it is NOT what a float bicubic+antialias resize gives
        bic = torch.nn.functional.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].float(), size=(518, 518),
                                              mode="bicubic", antialias=True)[0].permute(1, 2, 0)
        assert float((bic - torch.from_numpy(got).float()).abs().max()) > 2.0


def test_view_boxes_and_coords_equal_the_oracle():
    from dvt_amd import views as V
    boxes, coords = V.sample_view_boxes(40, (518, 518), 37, 37, np.random.RandomState(3))
    img = torch.zeros(3, 518, 518)
    oboxes, _, ocoords = oviews.make_views(img, 40, (518, 518), 37, 37, np.random.RandomState(3))
    assert np.array_equal(boxes, oboxes) and torch.equal(coords, ocoords)
    assert boxes[-1].tolist() == [0, 0, 518, 518, 0]
    assert float(coords.min()) >= 0.0 and float(coords.max()) <= 1.0


def test_stage1_outputs_roundtrip_through_the_stage2_reader(tmp_path):
    from dvt_amd.utils import misc
    model = "vit_base_patch14_dinov2.lvd142m"
    args = Namespace(save_root=str(tmp_path / "feats"), model=model, data_root=str(tmp_path / "VOC"))
    lines = ["VOC2007/JPEGImages/000005.jpg", "VOC2012/JPEGImages/2008_000008.jpg extra", "VOC2012/x/missing.png"]
    rng = np.random.RandomState(0)
    written = {}
    for line in lines[:2]:
        rel = line.split(" ")[0]
        fn = os.path.join(args.data_root, rel)
        raw, den = rng.randn(37, 37, 768).astype(np.float32), rng.randn(1, 37, 37, 768).astype(np.float32)
        raw_p, den_p = misc.output_paths(args.save_root, model, args.data_root, fn)
        misc.atomic_save_npy(raw_p, raw)   # [H, W, C]    main_img_denoising.py:144
        misc.atomic_save_npy(den_p, den)   # [1, H, W, C] main_img_denoising.py:145
        assert misc.check_if_file_exists(args, fn)
        written[rel] = (raw, den)
    feat_root = f"{args.save_root}/denoised_features/{model}"  # what main_denoiser.py is pointed at
    for line in lines[:2]:
        pair = stage2_reader.read_pair(feat_root, line)
        raw, den = written[line.split(" ")[0]]
        assert pair["original_feats"].shape == pair["denoised_feats"].shape == (37, 37, 768)
        assert np.array_equal(pair["original_feats"], raw) and np.array_equal(pair["denoised_feats"], den[0])
        assert pair["denoised_feats"].dtype == np.float32
    assert stage2_reader.read_pair(feat_root, lines[2]) is None
    assert not [p for p in (tmp_path / "feats").rglob("*") if ".tmp." in p.name]  # atomic writes left nothing


def test_wrapper_requires_weights_unless_random_init_is_allowed():
    from dvt_amd.models import PretrainedViTWrapper
    os.environ.pop("DVT_VIT_CHECKPOINT", None)
    with pytest.raises(RuntimeError, match="checkpoint"):
        PretrainedViTWrapper("vit_base_patch14_dinov2.lvd142m", stride=14)
    with pytest.warns(UserWarning, match="RANDOM"):
        w = PretrainedViTWrapper("vit_small_patch14_dinov2.lvd142m", stride=14, allow_random_init=True)
    assert w.n_output_dims == 384 and w.num_blocks == 12


def test_wrapper_loads_a_timm_layout_checkpoint(tmp_path):
    from dvt_amd.models import MODEL_LIST, PretrainedViTWrapper
    from dvt_amd.vit import random_state_dict
    sd = random_state_dict(384, 12, 14, 1370, seed=5)
    path = tmp_path / "vits14.pth"
    torch.save(sd, path)
    w = PretrainedViTWrapper("vit_small_patch14_dinov2.lvd142m", stride=14, checkpoint_path=str(path))
    assert torch.equal(w._state_dict["pos_embed"], sd["pos_embed"])
    # the reference's list, entry for entry (dvt/models/vit_wrapper.py:15-56)
    assert len(MODEL_LIST) == 20 and "vit_base_patch16_clip_224.openai" in MODEL_LIST
    assert not any(m.startswith("samvit") for m in MODEL_LIST)


def test_stage1_flag_table_keeps_the_reference_defaults():
    """EVERY flag of the reference's argparse table (main_img_denoising.py:152-208) with its default, plus
    this build's extras, whose defaults must not change what the reference's flags mean: --fp32_matmul highest (exact
    fp32), --fit_batch 0 (auto), --extract_launch_views 0 (auto; the reference's --extract_bsz 32 is a DataLoader batch
    and keeps its name and default)."""
    from dvt_amd import stage1
    a = stage1.get_args([])
    ref = {"model": "vit_base_patch14_dinov2.lvd142m", "stride_size": 14, "layer_depth_ratio": 1.0,
           "img_path": "demo/assets/demo/cat.jpg", "dtype": "float32", "data_root": None, "save_root": None,
           "start_idx": 0, "num_imgs": 100, "num_views": 768, "num_iters": 25000, "warmup_iters": 2500, "n_levels": 16,
           "freeze_shared_artifacts_after": 0.5, "lr": 0.01, "min_lr": 0.001, "weight_decay": 1e-5, "extract_bsz": 32,
           "pixel_bsz": 2048, "output_dir": "./work_dirs/demo", "num_vis_samples": 5, "vis_freq": 100, "seed": 0}
    for k, v in ref.items():
        assert getattr(a, k) == v, (k, getattr(a, k), v)
    assert tuple(a.input_size) == (518, 518)
    # the table above is complete: every option string the reference's parser declares is in it (parsed from the
    # reference's source when it is present -- this container; the GPU box has no /root/reference)
    ref_src = "/root/reference/main_img_denoising.py"
    if os.path.exists(ref_src):
        import re
        flags = set(re.findall(r'"--([a-z_0-9]+)"', open(ref_src).read()))
        assert flags == set(ref) | {"input_size"}, flags ^ (set(ref) | {"input_size"})
    assert a.fp32_matmul == "highest" and a.fit_batch == 0 and a.extract_launch_views == 0
    assert stage1.get_args(["--dtype", "float32", "--fp32_matmul", "high"]).fp32_matmul == "high"
    import pytest
    with pytest.raises(SystemExit):
        stage1.get_args(["--fp32_matmul", "medium"])


def test_launch_views_planning():
    """Views per extractor launch.  Round 3: 769 views at the cap of 128 -> 7 equal launches of 110 (109).  Round 4: the
    split is tile-round aware -- 605 M panels (110 views) give the N = 768 GEMMs 7.09 -> 8 rounds of 256 tiles, 682 panels
    (124 views) 7.99 -- and chosen by a small dynamic program; the reference's --extract_bsz does not enter."""
    from dvt_amd.vit import balanced_launch_views, plan_launches
    assert balanced_launch_views(769, 128) == 110 and balanced_launch_views(769, 32) == 31 and balanced_launch_views(1, 128) == 1
    plan = plan_launches(769, 128)
    assert sum(plan) == 769 and max(plan) <= 128 and plan == sorted(plan, reverse=True)
    assert plan[0] == 124, plan   # 682 M panels: 23.98 / 31.97 / 7.99 / 7.99 rounds for qkv / fc1 / proj / fc2

    def rounds(v, nt):
        return -(-(-(-v * 1408 // 256)) * nt // 256)
    for nt in (9, 12, 3):  # never more tile rounds than the equal split
        assert sum(rounds(v, nt) for v in plan) <= 6 * rounds(110, nt) + rounds(109, nt), (nt, plan)
    assert plan_launches(17, 128) == [17] and plan_launches(1, 128) == [1]
    assert sum(plan_launches(769, 32)) == 769 and max(plan_launches(769, 32)) <= 32
    os.environ["DVT_VIT_BALANCE"] = "1"
    try:
        assert plan_launches(769, 128) == [110] * 6 + [109]
    finally:
        del os.environ["DVT_VIT_BALANCE"]


def test_stage1_rejects_unknown_fp32_matmul_before_allocating():
    """Programmatic callers bypass argparse's `choices`: the driver itself refuses an unknown --fp32_matmul with the
    library's error type (the raise used an un-imported name), before any device allocation."""
    from dvt_amd import _lib, stage1
    with pytest.raises(_lib.DvtError, match="fp32_matmul"):
        stage1.Stage1(Namespace(fp32_matmul="medium"), "cpu")


def test_pipe_timeline_tool_on_a_synthetic_trace(tmp_path, capsys):
    """tools/pipe_timeline.py (the two-stream pipeline's timeline from a rocprofv3 kernel trace) on a hand-made `kernels`
    table: two extractor launches per image, a fit that runs beside the second image, one idle gap."""
    import importlib.util
    import sqlite3
    db = tmp_path / "t.db"
    c = sqlite3.connect(db)
    c.execute("create table kernels (name text, start integer, end integer, grid_x integer)")
    rows, t = [], 1000
    for img in range(3):
        for launch in range(2):
            rows.append(("im2col_kernel(float const*)", t, t + 100, 64)); t += 100
            for blk in range(2):
                rows.append(("void gemm_bf16_kernel_8p<1>(GemmBArgs)", t, t + 2000, 512)); t += 2000
                rows.append(("void attention_kernel_v2<15>(...)", t, t + 3000, 1024)); t += 3000
        if img == 0:
            t += 2000000  # the extractor waits once (2 ms)
    f = 1000 + 10200 + 2000000
    for step in range(40):
        rows.append(("void fit_rows_kernel<768, false, 16, 8>(FusedArgs)", f, f + 60, 128)); f += 70
        rows.append(("fit_backward_kernel<16>(BackwardArgs)", f, f + 40, 128)); f += 50
        rows.append(("void adam_dense_lazy_shadow_kernel<false>(AdamKArgs)", f, f + 30, 256)); f += 40
    c.executemany("insert into kernels values (?, ?, ?, ?)", rows)
    c.commit()
    c.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pipe_timeline", os.path.join(root, "tools", "pipe_timeline.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(str(db))
    mod.interference(str(db))
    out = capsys.readouterr().out
    assert "6 extractor launches -> 3 images" in out
    assert "idle before next image" in out and "between image 0 and 1" in out
    assert "extractor launch duration vs fit steps completed inside it" in out


def test_explicit_extract_bsz_bounds_the_launches():
    """`--extract_bsz` keeps the reference's default (32) in the table and is otherwise ignored -- the views per
    extractor launch are `--extract_launch_views` (default cap 400) -- but a value the user TYPED (to bound memory) bounds the
    launches when `--extract_launch_views` is left at 0."""
    from dvt_amd import stage1

    def cap(argv):
        a = stage1.get_args(argv)
        elv = int(a.extract_launch_views or 0)
        ebs = a.extract_bsz if getattr(a, "extract_bsz_explicit", False) else None
        return a.extract_bsz, max(1, elv if elv > 0 else (min(400, int(ebs)) if ebs else 400))

    assert cap([]) == (32, 400)
    assert cap(["--extract_bsz", "32"]) == (32, 32)
    assert cap(["--extract_bsz", "64"]) == (64, 64)
    assert cap(["--extract_bsz", "1000"]) == (1000, 400)
    assert cap(["--extract_bsz", "16", "--extract_launch_views", "128"]) == (16, 128)


def test_launch_views_planning_fp32_tiles():
    """Round 5: the exact-fp32 extractor's GEMMs run 128 x 128 tiles on 512 workgroup slots (two per CU); its launches are
    planned for THAT geometry below a cap of 160 views (8 GB of fp32 scratch for ViT-B): 769 views -> five launches whose
    N = 768 GEMMs fill whole rounds (at the round-4 cap of 32 views they ran 4.1 -> 5 rounds, 18 % idle)."""
    from dvt_amd.vit import plan_launches
    plan = plan_launches(769, 160, 1408, 768, 3072, fp32=True)
    assert sum(plan) == 769 and max(plan) <= 160 and len(plan) == 5, plan
    for v in plan:  # proj / fc2: M tiles x 6 N tiles over 512 slots: the last round at least three quarters full
        tiles = -(-v * 1408 // 128) * 6
        assert (tiles % 512 == 0) or (tiles % 512) / 512 >= 0.2, (v, tiles % 512)
    assert plan_launches(64, 160, 1408, 768, 3072, fp32=True) == [64]
    assert plan_launches(769, 400) == [398, 371]  # the bf16 plan is untouched


def test_fit_group_plan_tapers_the_tail():
    """Stage1.group_plan (round 6): groups of `fit_batch` images share every fit launch, but a group's fits start only when its
    last image is extracted -- so the LAST groups of a run of known length taper (..., 4, 4, 2, 1, 1) and only one fit drains
    with nothing beside it.  Every image is in exactly one group, order preserved, no group larger than the batch."""
    from dvt_amd.stage1 import Stage1
    for total in range(0, 40):
        for kb in (1, 2, 3, 4, 8):
            plan = Stage1.group_plan(total, kb)
            if kb <= 1:
                assert plan is None
                continue
            assert sum(plan) == total and all(1 <= g <= kb for g in plan), (total, kb, plan)
            if total > 4:
                assert plan[-1] == 1 and plan[-2] == 1, (total, kb, plan)
    assert Stage1.group_plan(20, 4) == [4, 4, 4, 4, 2, 1, 1]
    assert Stage1.group_plan(21, 4) == [1, 4, 4, 4, 4, 2, 1, 1]
    assert Stage1.group_plan(3, 4) == [1, 1, 1] and Stage1.group_plan(None, 4) is None
    # ... and only where the extractor sets the pace (Stage1.taper_pays / taper_model): the metric's configuration yes; the
    # reference's literal defaults (25 000 iterations: 5 s of fp32 fit against 1.9 s of extraction per image) no -- there a
    # 2 + 1 + 1 tail cost 5 s of a 22-s run of four images (profiles/r06/literal_defaults/)
    assert Stage1.group_plan(20, 4, taper=False) is None
    assert Stage1.taper_model(1000, 769, "bfloat16", "highest", 12, 768, 1370)
    assert Stage1.taper_model(1000, 769, "float32", "highest", 12, 768, 1370)
    assert not Stage1.taper_model(25000, 769, "float32", "highest", 12, 768, 1370)
    assert not Stage1.taper_model(25000, 769, "bfloat16", "highest", 12, 768, 1370)
    assert Stage1.taper_model(1000, 769, "bfloat16", "highest", 24, 1024, 1370)
