"""`Denoiser` -- the stage-2 generalizable denoiser behind the reference's class
(dvt/models/online_denoiser.py:13-104): same constructor arguments, same `forward` keywords and result keys,
same state-dict names (`pos_embed`, `denoiser.norm1.weight`, `denoiser.attn.qkv.weight`, ...; `denoiser.<i>.`
for num_blocks > 1), so checkpoints written by main_denoiser.py:239-251 (`{"denoiser": state_dict}`) load
unchanged.  Every parameter is a view into ONE flat device arena owned by `dvt_amd.s2.Stage2Engine`; compute is
the HIP library (csrc/dvt_stage2.hip), there is no CPU path.

Training differs from the reference in one place: the reference differentiates through autograd
(`loss.backward()`, main_denoiser.py:218-221); here `training_step(original_feats, denoised_feats)` runs
forward + loss + backward natively and leaves the gradients in `engine.grads` for the optimizer step.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import s2
from ..vit import resample_pos_embed
from .vit_wrapper import PretrainedViTWrapper


class _Holder(nn.Module):
    """A parameter container with the attribute path of the timm module it stands for."""


def _attach(root: nn.Module, dotted: str, p: nn.Parameter) -> None:
    parts = dotted.split(".")
    m = root
    for name in parts[:-1]:
        if not hasattr(m, name):
            m.add_module(name, _Holder())
        m = getattr(m, name)
    m.register_parameter(parts[-1], p)


class Denoiser(nn.Module):
    RESIZED_CACHE = 4  # resized inference copies kept (LRU)
    def __init__(self, noise_map_height: int = 37, noise_map_width: int = 37, feat_dim: int = 768,
                 vit: PretrainedViTWrapper = None, enable_pe: bool = True, num_blocks: int = 1,
                 device: torch.device | str = "cuda", seed: int | None = None):
        super().__init__()
        self.vit = vit
        self.noise_map_size = (noise_map_height, noise_map_width)
        cfg = s2.make_config(feat_dim, noise_map_height * noise_map_width, num_blocks, enable_pe)
        self.engine = s2.Stage2Engine(cfg, torch.device(device))
        g = None if seed is None else torch.Generator().manual_seed(seed)
        self.engine.init_parameters(g)
        # registration order of the reference: denoiser blocks, then pos_embed (online_denoiser.py:24-57)
        self.denoiser = _Holder()
        self.pos_embed = None
        views = self.engine.views()
        for name, v in views.items():
            if name != "pos_embed":
                _attach(self, name, nn.Parameter(v))
        if enable_pe:
            self.pos_embed = nn.Parameter(views["pos_embed"])
        if self.vit is not None:
            for p in self.vit.parameters():
                p.requires_grad = False
        self._resized = {}
        self._resized_version = -1

    # parameters are views of the engine's arena: moving the module must move the arena, not the views
    def _apply(self, fn, recurse=True):
        probe = fn(torch.empty(0, device=self.engine.device))
        if probe.device != self.engine.device or probe.dtype != torch.float32:
            raise NotImplementedError("Denoiser lives on the HIP device it was built on, in fp32 "
                                      "(construct it with device=...)")
        return self

    def load_state_dict(self, state_dict, strict: bool = True):
        sd = {k: v for k, v in state_dict.items() if not k.startswith("vit.")}  # main_denoiser.py:241-245
        self.engine.load_named(sd)
        self._resized.clear()
        return nn.modules.module._IncompatibleKeys([], [k for k in sd if k not in self.engine.layout])

    def _engine_for(self, h: int, w: int) -> s2.Stage2Engine:
        """Inference at another token grid: timm's `resample_abs_pos_embed` (online_denoiser.py:89) applied
        once to a copy of the parameters."""
        if (h, w) == self.noise_map_size or self.pos_embed is None and h * w == self.engine.cfg.tokens:
            return self.engine
        # resized copies: parameters only (no gradient / moment arenas), at most RESIZED_CACHE of them (LRU: dense-task
        # evaluation over variable-size inputs must not grow device memory without bound), and valid for ONE parameter
        # version -- any write through the engine (adamw_step, load_named) invalidates them, not only training_step
        if self._resized_version != self.engine.param_version:
            self._resized.clear()
            self._resized_version = self.engine.param_version
        if (h, w) in self._resized:
            self._resized[(h, w)] = self._resized.pop((h, w))  # most recently used last
        if (h, w) not in self._resized:
            while len(self._resized) >= self.RESIZED_CACHE:
                self._resized.pop(next(iter(self._resized)))
            c = self.engine.cfg
            e = s2.Stage2Engine(s2.make_config(c.dim, h * w, c.n_blocks, bool(c.enable_pe)), self.engine.device,
                                inference_only=True)
            src, dst = self.engine.views(), e.views()
            for k in dst:
                if k == "pos_embed":
                    gh, gw = self.noise_map_size
                    dst[k].copy_(resample_pos_embed(src[k].detach().cpu().reshape(1, gh * gw, c.dim), (h, w), 0)
                                 .to(e.device))
                else:
                    dst[k].copy_(src[k])
            self._resized[(h, w)] = e
        return self._resized[(h, w)]

    def forward(self, x, return_dict=False, return_channel_first=False, return_class_token=False, norm=True):
        class_tokens = None
        if self.vit is not None:
            if return_class_token:
                raise NotImplementedError("class tokens are not produced by the HIP extractor")
            with torch.no_grad():
                vit_outputs = self.vit.get_intermediate_layers(x, n=[self.vit.last_layer_index], norm=norm)
                original_feats = vit_outputs[0].permute(0, 2, 3, 1)
                x = original_feats
        else:
            original_feats = x
        b, h, w, c = x.shape
        eng = self._engine_for(h, w)
        with torch.no_grad():
            y = eng.forward(x.reshape(b, h * w, c).contiguous().float()).reshape(b, h, w, c)
        if return_channel_first:
            y = y.permute(0, 3, 1, 2)
        if return_dict:
            return {"denoised_feats": y, "original_feats": original_feats.detach(), "class_tokens": class_tokens}
        return y

    def training_step(self, original_feats: torch.Tensor, denoised_feats: torch.Tensor,
                      pred: torch.Tensor | None = None) -> torch.Tensor:
        """main_denoiser.py:212-220 in one native call; -> device tensor [loss, l2_loss, cosine_similarity_loss, 0]."""
        b, h, w, c = original_feats.shape
        if (h, w) != self.noise_map_size:
            raise ValueError("training runs at the noise-map resolution the model was built for")
        self._resized.clear()  # parameters are about to change
        return self.engine.train_step(original_feats.reshape(b, h * w, c), denoised_feats.reshape(b, h * w, c),
                                      None if pred is None else pred.reshape(b, h * w, c))
