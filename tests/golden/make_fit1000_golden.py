"""BASELINE.json configs[1] / configs[2] fit schedule as committed oracle fixtures.

The CPU oracle loop (oracle/fit.py == reference main_img_denoising.py:28-149, `denoise_an_image`) run
ONCE in the build container at the schedule the headline metric is quoted on:

    1000 Adam steps, warm-up 100, B = 2048 sampled rows per step,
    L = 16 levels / F = 8 / 2^20 hash (19.7 M grid parameters), 37 x 37 lattice, 64 views + the original,
    C = 768 (ViT-B/14, MLP 128 -> 384 -> 768)   -> tests/golden/fit1000_c768.npz
    C = 1024 (ViT-L/14, MLP 128 -> 512 -> 1024) -> tests/golden/fit1000_c1024.npz

    python tests/golden/make_fit1000_golden.py [768|1024]        (~6 min per configuration on 8 cores)
    python tests/golden/make_fit1000_golden.py 768:769           -> tests/golden/fit1000_c768_v769.npz: the metric's LITERAL
        configuration -- 768 views + the original = 1 052 761 rows, a 3.2-GB feature store, row indices beyond 2^20
        (main_img_denoising.py:64-76); the oracle's cost is per step (dense Adam), not per view

Inputs are the structured synthetic features of SURVEY.md 8d (tests/test_gpu_fit.synthetic_image: smooth
field of the global coordinates + a lattice artefact shared by all views + noise), regenerated from seeds on
the GPU box; checksums guard that the torch CPU generator reproduces them.  Stored: per-step scalars of ALL
1000 steps, the saved tensor `denoised_feats` (main_img_denoising.py:121-130) as fp16, and -- as the measured
noise floor of a 1000-step run -- the same oracle run again with every initial parameter perturbed by 1e-6
relative (a stand-in for fp reassociation): its per-step losses and its per-patch cosine against
the unperturbed run.  tests/test_gpu_parity_full.py::test_fit_baseline_schedule_vs_oracle_fixture consumes it.
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

V, H, T, WARM, B = 65, 37, 1000, 100, 2048
KEYS = ("loss", "patch_l2_loss", "cosine_similarity_loss", "residual_loss", "residual_sparsity_loss")
DATA_SEED, INIT_SEED, IDX_SEED = 31, 0, 17


def out_path(C, views=V):
    return os.path.join(ROOT, "tests", "golden", f"fit1000_c{C}.npz" if views == V else f"fit1000_c{C}_v{views}.npz")


def checksum(tensors) -> float:
    return float(sum(float(t.detach().double().abs().sum()) for t in tensors))


def inputs(C, views=V):
    """(feats [views,H,H,C], xy [views,H,H,2], idx [T,B] int32) -- identical on the GPU box (seeded CPU generators)."""
    from tests.test_gpu_fit import synthetic_image
    feats, xy = synthetic_image(views, H, H, C, seed=DATA_SEED + C)
    idx = np.random.RandomState(IDX_SEED).randint(0, views * H * H, (T, B)).astype(np.int32)
    return feats, xy, idx


def fresh_modules(C, seed=INIT_SEED):
    from oracle.models import NeuralFeatureFieldOracle, SingleImageDenoiserOracle
    torch.manual_seed(seed)
    d = SingleImageDenoiserOracle(H, H, C, 11)          # main_img_denoising.py:39-44
    f = NeuralFeatureFieldOracle(feat_dim=C, n_levels=16)  # :46
    return d, f


def run(C, views=V):
    from oracle import fit as ofit
    feats, xy, idx = inputs(C, views)

    def one(perturb):
        d, f = fresh_modules(C)
        if perturb:
            g = torch.Generator().manual_seed(99)
            with torch.no_grad():
                for p in list(d.parameters()) + list(f.parameters()):
                    p.mul_(1.0 + perturb * torch.randn(p.shape, generator=g))
        init = checksum(list(d.parameters()) + list(f.parameters()))
        t0 = time.time()
        logs = ofit.fit_image(d, f, feats, xy, idx, num_iters=T, warmup_iters=WARM, log_every=1)
        print(f"C={C} perturb={perturb}: {T} oracle steps in {time.time() - t0:.0f} s", flush=True)
        den = ofit.final_denoised_feats(d, f, feats, xy)[0]
        tab = np.array([[logs[s].get(k, 0.0) for k in KEYS] for s in range(T)], np.float64)
        return init, tab, den

    init, tab, den = one(0.0)
    _, tab_p, den_p = one(1e-6)
    cos = torch.nn.functional.cosine_similarity(den.reshape(-1, C).double(), den_p.reshape(-1, C).double(), dim=-1)
    np.savez_compressed(
        out_path(C, views), losses=tab, denoised_f16=den.numpy().astype(np.float16), losses_perturbed=tab_p,
        perturbed_cos=np.array([float(cos.mean()), float(cos.min())]),
        feats_checksum=np.float64(checksum([feats, xy])), init_checksum=np.float64(init),
        meta=np.array([views, H, T, WARM, B, C, DATA_SEED + C, INIT_SEED, IDX_SEED], np.int64))
    rel = np.abs(tab_p[:, 0] - tab[:, 0]) / np.abs(tab[:, 0])
    print(f"wrote {out_path(C, views)} ({os.path.getsize(out_path(C, views)) / 1e6:.2f} MB); loss {tab[0, 0]:.4f} -> {tab[-1, 0]:.4f}; "
          f"oracle vs itself (init perturbed 1e-6): loss rel diff max {rel.max():.2e} (last step {rel[-1]:.2e}), "
          f"saved tensor cos mean {cos.mean():.6f} min {cos.min():.6f}")


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("ORACLE_THREADS", os.cpu_count() or 8)))
    for a in (sys.argv[1:] or ["768", "1024"]):  # "768" or "768:769" = C[:views]
        c, _, v = a.partition(":")
        run(int(c), int(v) if v else V)
