"""Generate the golden fixtures in tests/golden/ by IMPORTING THE REFERENCE
(/root/reference, available only in the build container; fixtures are committed because the
reference cannot travel to the GPU box).

    python tests/golden/make_golden.py

Pins: SingleImageDenoiser.forward (dvt/models/offline_denoiser.py:62-171, both phases +
visualization outputs, incl. autograd gradients), misc.adjust_learning_rate
(dvt/utils/misc.py:306-322), make_patch_coordinates (main_img_denoising.py:21-25 restated
there because the driver module cannot be imported without timm/torchvision) and the numpy
index stream (main_img_denoising.py:73 with fix_random_seeds(0)).
The hash grid (tiny-cuda-nn) and the ViT (timm) are third-party and absent: not pinned here.
"""
import importlib.util
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    pkg = types.ModuleType("dvt_ref_models")
    pkg.__path__ = [os.path.join(REF, "dvt", "models")]
    sys.modules["dvt_ref_models"] = pkg
    spec = importlib.util.spec_from_file_location(
        "dvt_ref_models.offline_denoiser", os.path.join(REF, "dvt/models/offline_denoiser.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    spec2 = importlib.util.spec_from_file_location("dvt_ref_misc",
                                                   os.path.join(REF, "dvt/utils/misc.py"))
    misc = importlib.util.module_from_spec(spec2)
    spec2.loader.exec_module(misc)
    return mod.SingleImageDenoiser, misc


class FieldStub(torch.nn.Module):
    """A tiny deterministic stand-in for NeuralFeatureField (the real one needs tcnn)."""

    def __init__(self, c):
        super().__init__()
        self.lin = torch.nn.Linear(2, c)

    def forward(self, xy):
        return torch.sin(self.lin(xy) * 3.0)


def denoiser_case(SID, phase2: bool, seed: int):
    torch.manual_seed(seed)
    C, H, W, n = 32, 5, 6, 48
    den = SID(noise_map_height=H, noise_map_width=W, feat_dim=C, layer_index=3)
    field = FieldStub(C)
    raw = torch.randn(n, C) * 2.0
    xy = torch.rand(n, 2)
    # lattice coords exactly as the driver builds them (main_img_denoising.py:21-25, :58)
    py, px = torch.linspace(-1, 1, H), torch.linspace(-1, 1, W)
    gy, gx = torch.meshgrid(py, px, indexing="ij")
    lattice = torch.stack([gx, gy], -1).reshape(-1, 2)
    pos = torch.randint(0, H * W, (n,))
    sac = lattice[pos]
    if phase2:
        den.stop_shared_artifacts_grad()
        den.start_residual_predictor()
    out = den(raw_vit_outputs=raw, global_pixel_coords=xy, neural_field=field,
              shared_artifact_coords=sac)
    (out["loss"] * 1024.0).backward()
    d = {
        "raw": raw, "xy": xy, "sac": sac, "pos": pos, "G": den.shared_artifacts.detach(),
        "field_w": field.lin.weight.detach(), "field_b": field.lin.bias.detach(),
        "g_field_w": field.lin.weight.grad, "g_field_b": field.lin.bias.grad,
    }
    for k, v in den.residual_predictor.state_dict().items():
        d["rp." + k] = v
    for k, v in out.items():
        d["out." + k] = v.detach()
    if not phase2:
        d["g_G"] = den.shared_artifacts.grad
    else:
        for name, p in den.residual_predictor.named_parameters():
            d["g_rp." + name] = p.grad
    # inference/visualization call (main_img_denoising.py:124-129)
    rawv = torch.randn(1, H, W, C)
    xyv = torch.rand(1, H, W, 2)
    with torch.no_grad():
        vis = den(raw_vit_outputs=rawv, global_pixel_coords=xyv, neural_field=field,
                  return_visualization=True)
    d["vis.raw"], d["vis.xy"] = rawv, xyv
    for k, v in vis.items():
        d["vis." + k] = v.detach()
    return {k: v.numpy() for k, v in d.items()}


def main():
    SID, misc = load_reference()
    np.savez(os.path.join(OUT, "denoiser_phase1.npz"), **denoiser_case(SID, False, 0))
    np.savez(os.path.join(OUT, "denoiser_phase2.npz"), **denoiser_case(SID, True, 1))

    # LR schedule through the reference's own function
    class Opt:
        param_groups = [{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.5}]

    rows = []
    for (lr, min_lr, warm, iters) in [(0.01, 0.001, 2500, 25000), (0.01, 0.001, 100, 1000),
                                      (0.01, 0.001, 2500, 1000)]:
        args = Namespace(lr=lr, min_lr=min_lr, warmup_iters=warm, num_iters=iters)
        for step in sorted({0, 1, 50, 99, 100, 101, 499, 500, 501, 999, 2499, 2500, 2501, 12500,
                            24999} & set(range(iters))):
            v = misc.adjust_learning_rate(Opt, step, args)
            rows.append((lr, min_lr, warm, iters, step, v, Opt.param_groups[1]["lr"]))
    np.save(os.path.join(OUT, "lr_schedule.npy"), np.asarray(rows, np.float64))

    # index stream: fix_random_seeds(0) then per-step np.random.randint(0, N, B)
    misc.fix_random_seeds(0)
    N, B, T = 1052761, 2048, 4
    stream = np.stack([np.random.randint(0, N, B) for _ in range(T)])
    np.save(os.path.join(OUT, "index_stream_seed0.npy"), stream.astype(np.int32))
    print("wrote fixtures to", OUT)


if __name__ == "__main__":
    main()
