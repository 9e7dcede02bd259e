"""GPU: stage-2 (N3) parity through the C ABI -- `Denoiser` forward, the fused forward+loss+backward step, AdamW
and a short training run against the CPU oracle (oracle/stage2.py: timm-Block restatement + torch autograd +
torch.optim.AdamW) on identical parameters and batches.  fp32 on both sides; tolerances are relative L2."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import stage2 as O
from oracle import vit as OV

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def make_pair(h, w, c, blocks, seed=0, enable_pe=True):
    from dvt_amd.models import Denoiser
    torch.manual_seed(seed)
    ref = O.Denoiser(h, w, c, enable_pe, blocks)
    with torch.no_grad():  # non-trivial LayerNorm parameters so that their gradients are exercised
        for n, p in ref.named_parameters():
            if "norm" in n:
                p.add_(0.1 * torch.randn_like(p))
    mine = Denoiser(h, w, c, None, enable_pe, blocks, device=DEV)
    mine.load_state_dict(ref.state_dict())
    return ref, mine


@pytest.mark.parametrize("h,w,c,blocks,batch", [(7, 7, 384, 1, 3), (5, 9, 384, 2, 2), (8, 8, 1024, 1, 1)])
def test_forward_vs_oracle(h, w, c, blocks, batch):
    ref, mine = make_pair(h, w, c, blocks)
    assert list(mine.state_dict()) == list(ref.state_dict())  # same names, same order
    x = torch.randn(batch, h, w, c)
    with torch.no_grad():
        want = ref(x)
    got = mine(x.to(DEV))
    assert got.shape == want.shape and rel(got, want) < 2e-6, rel(got, want)
    d = mine(x.to(DEV), return_dict=True, return_channel_first=True)
    assert d["denoised_feats"].shape == (batch, c, h, w) and d["class_tokens"] is None
    assert torch.equal(d["original_feats"].cpu(), x)


# (11 x 11 = 121 tokens -> 128 padded rows: the fused attention-rows kernel and the 128-row GEMM tiles at other widths / depths;
# 7 x 7, 6 x 6: 64 padded rows, the 64 x 64-tile fallbacks)
@pytest.mark.parametrize("h,w,c,blocks,batch", [(7, 7, 384, 1, 3), (6, 6, 384, 2, 2), (37, 37, 768, 1, 2), (11, 11, 384, 2, 3),
                                                (11, 11, 1024, 1, 2)])
def test_step_gradients_vs_autograd(h, w, c, blocks, batch):
    ref, mine = make_pair(h, w, c, blocks, seed=1)
    torch.manual_seed(2)
    x, t = torch.randn(batch, h, w, c), torch.randn(batch, h, w, c)
    loss, l2, cos = O.loss_fn(ref(x), t)
    loss.backward()
    pred = torch.empty(batch, h, w, c, device=DEV)
    got = mine.training_step(x.to(DEV), t.to(DEV), pred).cpu()
    assert abs(got[0] - loss.item()) < 2e-6 * abs(loss.item()) + 1e-7
    assert abs(got[1] - l2.item()) < 2e-6 * l2.item() and abs(got[2] - cos.item()) < 1e-6
    with torch.no_grad():
        assert rel(pred, ref(x)) < 2e-6
    grads = mine.engine.views(mine.engine.grads)
    worst = 0.0
    for n, p in ref.named_parameters():
        r = rel(grads[n], p.grad)
        worst = max(worst, r)
        assert r < 2e-5, (n, r)
    print(f"[stage-2 step {h}x{w}x{c}, {blocks} block(s)] loss {got[0]:.6f} (oracle {loss.item():.6f}); worst "
          f"gradient rel-L2 {worst:.2e}")
    # a second call ACCUMULATES
    mine.training_step(x.to(DEV), t.to(DEV))
    assert rel(grads["denoiser.attn.qkv.weight" if blocks == 1 else "denoiser.0.attn.qkv.weight"],
               2 * dict(ref.named_parameters())["denoiser.attn.qkv.weight" if blocks == 1 else
                                                "denoiser.0.attn.qkv.weight"].grad) < 2e-5


def test_step_kernel_choices_agree(built_lib):
    """Round 6: the step at the metric's shape (37 x 37 x 768) under every kernel choice of dvt_tune_set(18, mask) -- 63 = default:
    linear layers' forward / data- / weight-gradient GEMMs on the 128 x 128 tile, softmax fused into the attention products
    (s2_attn_rows_kernel), weight gradients on a side stream beside the data gradients; 0 = round 5's flow (64 x 64 tile, GEMM + softmax passes) -- gives the same loss and gradients up to
    summation order; each choice is also held against autograd by test_step_gradients_vs_autograd under the default."""
    ref, mine = make_pair(37, 37, 768, 1, seed=3)
    torch.manual_seed(4)
    x, t = torch.randn(2, 37, 37, 768, device=DEV), torch.randn(2, 37, 37, 768, device=DEV)
    out = {}
    try:
        for mask in (63, 0, 7, 24, 16, 31, 32):
            assert built_lib.dvt_tune_set(18, mask) == 0
            mine.engine.grads.zero_()
            loss = mine.training_step(x, t).cpu().clone()
            out[mask] = (loss, mine.engine.grads.clone())
    finally:
        assert built_lib.dvt_tune_set(18, 63) == 0
    assert built_lib.dvt_tune_set(18, 64) == -1
    base_loss, base_g = out[0]
    for mask, (loss, g) in out.items():
        assert abs(float(loss[0] - base_loss[0])) < 1e-6 * abs(float(base_loss[0])), mask
        r = rel(g, base_g)
        print(f"[stage-2 kernel choices] mask {mask:2d}: gradient arena rel-L2 vs round 5's flow {r:.2e}")
        assert r < 5e-6, (mask, r)
    assert not torch.equal(out[63][1], base_g)  # (the choices really are different kernels)


def test_adamw_vs_torch():
    from dvt_amd import _lib
    import ctypes as C
    torch.manual_seed(0)
    n = 4096 * 3
    p0 = torch.randn(n)
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p_ref], betas=(0.9, 0.999), weight_decay=1e-2)
    p, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 8):
        g = torch.randn(n) * (10.0 ** (step % 3 - 1))
        lr = 1e-3 * step
        for grp in opt.param_groups:
            grp["lr"] = lr
        p_ref.grad = g.clone()
        opt.step()
        gd = (2.0 * g).to(DEV)  # the kernel halves it again (grad_scale = 1 / world)
        _lib.check(_lib.lib().dvt_adamw_step(_lib.ptr(p), _lib.ptr(gd), _lib.ptr(m), _lib.ptr(v), n, lr, 0.9, 0.999,
                                             1e-8, 1e-2, step, 0.5, _lib.stream()), "dvt_adamw_step")
        assert float(gd.abs().max()) == 0.0  # fused zero_grad
        assert rel(p, p_ref) < 1e-6, (step, rel(p, p_ref))


def test_training_run_vs_oracle():
    """main_denoiser.py:204-221 for 12 steps: per-step losses and the final parameters."""
    h, w, c, batch, iters = 7, 7, 384, 4, 12
    ref, mine = make_pair(h, w, c, 1, seed=3)
    torch.manual_seed(4)
    data = [(torch.randn(batch, h, w, c), torch.randn(batch, h, w, c)) for _ in range(iters)]
    lr_base = O.scaled_lr(2e-4, 32, 8)
    log = O.train(ref, iter(data), iters, lr_base, 1e-6, 1e-5)
    from dvt_amd.stage2 import CosineScheduler
    sched = CosineScheduler(lr_base, 1e-6, iters, warmup_iters=int(iters * 0.15), start_warmup_value=0)
    losses = []
    for step, (x, t) in enumerate(data):
        losses.append(mine.training_step(x.to(DEV), t.to(DEV)).cpu()[0].item())
        mine.engine.adamw_step(float(sched[step]), 1e-5)
    want = [l[0] for l in log]
    assert np.allclose(losses, want, rtol=2e-5), (losses, want)
    sd = mine.state_dict()
    worst = max(rel(sd[k], v) for k, v in ref.state_dict().items())
    print(f"[stage-2 training, {iters} steps] loss {losses[0]:.5f} -> {losses[-1]:.5f} (oracle {want[0]:.5f} -> "
          f"{want[-1]:.5f}); worst parameter rel-L2 after training {worst:.2e}")
    assert worst < 1e-4


def test_pos_embed_resample_at_other_resolution():
    ref, mine = make_pair(7, 7, 384, 1, seed=5)
    x = torch.randn(2, 9, 9, 384)
    with torch.no_grad():
        pe = OV.resample_abs_pos_embed(ref.pos_embed, (9, 9), num_prefix_tokens=0)
        want = ref.denoiser(x.reshape(2, 81, 384) + pe).reshape(2, 9, 9, 384)
    got = mine(x.to(DEV))
    assert rel(got, want) < 2e-6


def test_cpu_tensors_are_refused():
    _, mine = make_pair(7, 7, 384, 1)
    with pytest.raises(Exception):
        mine.engine.forward(torch.randn(1, 49, 384))
    with pytest.raises(NotImplementedError):
        mine.to("cpu")


def test_stage2_driver_end_to_end(tmp_path):
    """Stage-1-layout files -> `python -m dvt_amd.stage2` loop -> checkpoints; losses against the oracle trainer fed
    the same sample order."""
    from dvt_amd import stage2
    root, n, bs, iters, H = str(tmp_path), 6, 2, 6, 5
    rng = np.random.default_rng(11)
    os.makedirs(f"{root}/denoised_features/m/")
    os.makedirs(f"{root}/raw_features/m/")
    for i in range(n):
        np.save(f"{root}/raw_features/m/im{i}.npy", rng.standard_normal((1, H, H, 384)).astype(np.float32))
        np.save(f"{root}/denoised_features/m/im{i}.npy", rng.standard_normal((H, H, 384)).astype(np.float32))
    with open(f"{root}/list.txt", "w") as f:
        f.write("".join(f"im{i}.jpg 0\n" for i in range(n)))
    args = stage2.get_args(["--model", "vit_small_patch14_dinov2.lvd142m", "--feat_root", f"{root}/denoised_features/m/",
                            "--data_list_path", f"{root}/list.txt", "--batch_size", str(bs), "--num_iterations",
                            str(iters), "--output_root", f"{root}/work", "--save_freq", "4", "--log_freq", "1",
                            "--num_workers", "2", "--input_size", str(14 * H)])
    out = stage2.train(args, 0, 1, DEV)
    ck = torch.load(f"{out['log_dir']}/checkpoints/latest.pth", weights_only=False)
    assert ck["step"] == iters - 1
    # oracle: same initial parameters are not available (the product draws its own init), so compare the LOSS CURVE
    # of an oracle started from the product's initial state, recovered from the step-0 checkpoint minus one update...
    # simpler and exact: restart both from the step-0 checkpoint's successor is not possible either; instead train
    # the oracle from the checkpointed step-4 state for the last step and compare that step's loss.
    ck4 = torch.load(f"{out['log_dir']}/checkpoints/ckpt_000004.pth", weights_only=False)
    ref = O.Denoiser(H, H, 384, True, 1)
    ref.load_state_dict(ck4["denoiser"])
    ds = stage2.PairedFeatureList(f"{root}/list.txt", f"{root}/denoised_features/m/")
    idx = [i % n for i in range(5 * bs, 6 * bs)]  # InfiniteSampler order, step 5
    x = torch.from_numpy(np.stack([ds[i][0] for i in idx]))
    t = torch.from_numpy(np.stack([ds[i][1] for i in idx]))
    with torch.no_grad():
        want = O.loss_fn(ref(x), t)[0].item()
    got = [h for h in out["history"] if h["step"] == 5][0]["loss"]
    assert abs(got - want) < 2e-5 * abs(want), (got, want)
    assert math.isfinite(out["history"][-1]["iter_time"])
