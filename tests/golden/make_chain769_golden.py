"""The WHOLE chain of BASELINE.json configs[1] at the metric's literal configuration, as one committed oracle fixture.

    python tests/golden/make_chain769_golden.py        (~10 min of oracle ViT + ~2 x 3 min of oracle fit on 8 cores)

Chain (reference main_img_denoising.py:309-352), every number the headline metric is quoted on:
    the demo image (the base-resized uint8 image committed in cat_demo.npz = /root/reference/demo/cat.jpg through
    :279-286) -> 768 random resized crops + the original (dvt/dataset/transform.py:39-76; boxes stored) -> fp32 oracle
    DINOv2 ViT-B/14 geometry, 12 blocks, 518 x 518 (vit_wrapper.py:122-143; random-init, seed-fixed, well-conditioned:
    no checkpoint can be fetched here) -> the 769 x 1369 = 1 052 761-row feature store -> denoise_an_image (:28-149):
    1000 Adam steps of 2048 rows, warm-up 100, L = 16 / F = 8 / 2^20 hash -> the saved tensor `denoised_feats` (:121-130).

Stored: the 769 boxes, `denoised_feats` as fp16, every step's loss scalars, the raw features of the original view (every
8th channel, fp16) and of view 0 (the same), checksums of the regenerated ViT weights / initial parameters, and the oracle
against ITSELF with every initial fit parameter perturbed by 1e-6 relative (the noise floor of a 1000-step run).
tests/test_gpu_parity_full.py::test_chain_metric_configuration_vs_oracle_fixture runs HIP view synthesis -> HIP ViT -> HIP
fit from the same weights / initial parameters / index stream against it.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "denoising-vit_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

OUT = os.path.join(ROOT, "tests", "golden", "chain769_c768.npz")
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
V, T, WARM, B, SIZE, HP, C = 768, 1000, 100, 2048, (518, 518), 37, 768
VIEW_SEED, INIT_SEED, IDX_SEED, VIT_SEED = 21, 0, 23, 0
KEYS = ("loss", "patch_l2_loss", "cosine_similarity_loss", "residual_loss", "residual_sparsity_loss")


def vit_weights():
    from dvt_amd.vit import random_state_dict
    return random_state_dict(C, 12, 14, 1370, seed=VIT_SEED, well_conditioned=True)


def fresh_modules(seed=INIT_SEED):
    from oracle.models import NeuralFeatureFieldOracle, SingleImageDenoiserOracle
    torch.manual_seed(seed)
    d = SingleImageDenoiserOracle(HP, HP, C, 11)          # main_img_denoising.py:39-44
    f = NeuralFeatureFieldOracle(feat_dim=C, n_levels=16)  # :46
    return d, f


def checksum(tensors) -> float:
    return float(sum(float(t.detach().double().abs().sum()) for t in tensors))


def index_stream():
    return np.random.RandomState(IDX_SEED).randint(0, (V + 1) * HP * HP, (T, B)).astype(np.int32)


def main():
    from oracle import fit as ofit
    from oracle import views as oviews
    from oracle import vit as ovit

    torch.set_num_threads(int(os.environ.get("ORACLE_THREADS", os.cpu_count() or 8)))
    img_u8 = np.load(os.path.join(ROOT, "tests", "golden", "cat_demo.npz"))["image_u8"]
    _, x = oviews.base_transform(img_u8, SIZE, MEAN, STD)
    boxes, views, coords = oviews.make_views(x, V, SIZE, HP, HP, np.random.RandomState(VIEW_SEED))
    sd = vit_weights()
    t0 = time.time()
    feats = torch.empty(V + 1, HP, HP, C)
    with torch.no_grad():
        for i in range(0, V + 1, 4):
            feats[i:i + 4] = ovit.forward_features(sd, views[i:i + 4], 14, 14)
            if i % 64 == 0:
                print(f"oracle ViT: view {i} at {time.time() - t0:.0f} s", flush=True)
    print(f"oracle ViT: {V + 1} views in {time.time() - t0:.0f} s", flush=True)
    del views
    idx = index_stream()

    def one(perturb):
        d, f = fresh_modules()
        if perturb:
            g = torch.Generator().manual_seed(99)
            with torch.no_grad():
                for p in list(d.parameters()) + list(f.parameters()):
                    p.mul_(1.0 + perturb * torch.randn(p.shape, generator=g))
        init = checksum(list(d.parameters()) + list(f.parameters()))
        t1 = time.time()
        logs = ofit.fit_image(d, f, feats, coords, idx, num_iters=T, warmup_iters=WARM, log_every=1)
        print(f"oracle fit (perturb {perturb}): {T} steps in {time.time() - t1:.0f} s", flush=True)
        den = ofit.final_denoised_feats(d, f, feats, coords)[0]
        return init, np.array([[logs[s].get(k, 0.0) for k in KEYS] for s in range(T)], np.float64), den

    init, tab, den = one(0.0)
    _, tab_p, den_p = one(1e-6)
    cos = torch.nn.functional.cosine_similarity(den.reshape(-1, C).double(), den_p.reshape(-1, C).double(), dim=-1)
    np.savez_compressed(
        OUT, boxes=boxes, denoised_f16=den.numpy().astype(np.float16), losses=tab, losses_perturbed=tab_p,
        perturbed_cos=np.array([float(cos.mean()), float(cos.min())]),
        raw_orig_f16_sub=feats[-1, :, :, ::8].numpy().astype(np.float16),
        raw_view0_f16_sub=feats[0, :, :, ::8].numpy().astype(np.float16),
        feats_abs_mean=np.float64(float(feats.abs().mean())),
        vit_checksum=np.float64(checksum(sd.values())), init_checksum=np.float64(init),
        meta=np.array([V, T, WARM, B, C, VIEW_SEED, INIT_SEED, IDX_SEED, VIT_SEED], np.int64))
    print(f"wrote {OUT} ({os.path.getsize(OUT) / 1e6:.2f} MB); loss {tab[0, 0]:.4f} -> {tab[-1, 0]:.4f}; oracle vs itself "
          f"(init perturbed 1e-6): saved tensor cos mean {cos.mean():.6f} min {cos.min():.6f}")


if __name__ == "__main__":
    main()
