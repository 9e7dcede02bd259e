"""CPU: the ViT oracle (oracle/vit.py, restating timm's DINOv2 forward) against the
independent `transformers` Dinov2Model carrying the same weights -- the second opinion
SURVEY.md 8c names, since timm itself is absent."""
import pytest
import torch

from oracle import vit as ovit


@pytest.mark.parametrize("dim,depth,img", [(128, 2, 56), (192, 3, 70)])
def test_oracle_vit_equals_hf_dinov2(dim, depth, img):
    pytest.importorskip("transformers")
    from dvt_amd.vit import random_state_dict
    g = img // 14
    sd = random_state_dict(dim, depth, 14, 1 + g * g, seed=3, well_conditioned=True)
    x = torch.randn(2, 3, img, img, generator=torch.Generator().manual_seed(0))
    mine = ovit.forward_features(sd, x, 14, 14)
    hf = ovit.to_hf_dinov2(sd, img, 14)
    with torch.no_grad():
        ref = hf(pixel_values=x).last_hidden_state[:, 1:].reshape(2, g, g, dim)
    assert mine.shape == ref.shape
    torch.testing.assert_close(mine, ref, rtol=1e-5, atol=1e-5)


def test_oracle_vit_with_registers_equals_hf():
    """the *_reg4_* layout (cls, 4 register tokens, patches; pos_embed on patches only) against the
    independent transformers Dinov2WithRegistersModel"""
    pytest.importorskip("transformers")
    from dvt_amd.vit import random_state_dict
    dim, depth, img, g = 128, 2, 56, 4
    sd = random_state_dict(dim, depth, 14, g * g, seed=5, well_conditioned=True, n_reg=4)
    assert sd["reg_token"].shape == (1, 4, dim) and sd["pos_embed"].shape == (1, g * g, dim)
    x = torch.randn(2, 3, img, img, generator=torch.Generator().manual_seed(1))
    mine = ovit.forward_features(sd, x, 14, 14)
    hf = ovit.to_hf_dinov2(sd, img, 14)
    with torch.no_grad():
        ref = hf(pixel_values=x).last_hidden_state[:, 5:].reshape(2, g, g, dim)
    torch.testing.assert_close(mine, ref, rtol=1e-5, atol=1e-5)


def test_pos_embed_resampling_host_equals_oracle_and_is_identity_on_same_grid():
    """stride override (vit_wrapper.py:78-91): 56 px, patch 14, stride 7 -> 7x7 tokens from a 4x4
    checkpoint grid.  The product's host-side resampler (dvt_amd.vit.resample_pos_embed) and the
    oracle's restatement of timm's resample_abs_pos_embed must agree; same grid = untouched."""
    from dvt_amd.vit import random_state_dict, resample_pos_embed
    sd = random_state_dict(128, 1, 14, 1 + 16, seed=2, well_conditioned=True)
    a = resample_pos_embed(sd["pos_embed"], (7, 7), 1)
    b = ovit.resample_abs_pos_embed(sd["pos_embed"], (7, 7), num_prefix_tokens=1)
    assert a.shape == (1, 50, 128) and torch.equal(a, b)
    assert torch.equal(a[:, 0], sd["pos_embed"][:, 0])  # the cls position is carried over
    assert resample_pos_embed(sd["pos_embed"], (4, 4), 1) is sd["pos_embed"]
    # bicubic up-sampling reproduces a constant field and keeps the value range plausible
    const = torch.ones(1, 17, 8)
    assert torch.allclose(resample_pos_embed(const, (9, 9), 1), torch.ones(1, 82, 8), atol=1e-6)
    x = torch.randn(1, 3, 56, 56)
    out = ovit.forward_features(sd, x, 14, 7)
    assert out.shape == (1, 7, 7, 128) and torch.isfinite(out).all()


def test_oracle_vit_intermediate_layer_and_stride():
    from dvt_amd.vit import random_state_dict
    sd = random_state_dict(128, 3, 14, 1 + 16, seed=0, well_conditioned=True)
    x = torch.randn(1, 3, 56, 56)
    a = ovit.forward_features(sd, x, 14, 14, n_blocks=2)
    b = ovit.forward_features(sd, x, 14, 14, n_blocks=3)
    assert a.shape == b.shape == (1, 4, 4, 128) and not torch.allclose(a, b)
    # layer_index = int(ratio * last_layer_index) (main_img_denoising.py:241)
    assert int(1.0 * 11) == 11 and int(0.5 * 11) == 5


def test_view_coords_restatement():
    """transform.py:55-73: crop EDGES (not patch centres), x mirrored on flip; the last sample
    is the full image with linspace(0,1) coords (main_img_denoising.py:337)."""
    import numpy as np
    from dvt_amd import views as V
    c = V.crop_coords(10, 20, 100, 200, 518, 518, 37, 37, flip=False)
    assert c.shape == (37, 37, 2)
    assert abs(float(c[0, 0, 0]) - 20 / 518) < 1e-7 and abs(float(c[0, 0, 1]) - 10 / 518) < 1e-7
    assert abs(float(c[-1, -1, 0]) - 220 / 518) < 1e-6 and abs(float(c[-1, -1, 1]) - 110 / 518) < 1e-6
    f = V.crop_coords(10, 20, 100, 200, 518, 518, 37, 37, flip=True)
    assert torch.allclose(f[:, :, 0], c[:, :, 0].flip(1), atol=1e-7) and torch.equal(f[:, :, 1], c[:, :, 1])
    rng = np.random.RandomState(0)
    boxes, coords = V.sample_view_boxes(50, (518, 518), 37, 37, rng)
    assert boxes.shape == (51, 5) and coords.shape == (51, 37, 37, 2)
    area = boxes[:-1, 2] * boxes[:-1, 3] / 518 ** 2
    assert area.min() > 0.09 and area.max() < 0.52  # scale=(0.1, 0.5) up to rounding
    ar = boxes[:-1, 3] / boxes[:-1, 2]
    assert ar.min() > 0.74 and ar.max() < 1.35
    assert float(coords.min()) >= 0 and float(coords.max()) <= 1
    assert torch.equal(coords[-1], V.make_patch_coordinates(37, 37, 0.0, 1.0))
