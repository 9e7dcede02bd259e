"""Host side of the stage-2 denoiser kernels (csrc/dvt_stage2.hip, C ABI in include/dvt_stage2.h).

`Stage2Engine` owns four flat fp32 arenas on the device -- parameters, gradients and the two AdamW moments,
all in the layout `dvt_s2_param_offsets` reports -- plus the activation workspace, and exposes the three
operations the reference's stage-2 loop consists of (main_denoiser.py:212-221): `forward`, `train_step`
(forward + loss + backward, gradients accumulated into the flat gradient arena) and `adamw_step`.  A
data-parallel trainer all-reduces `engine.grads` between the last two (one flat RCCL all-reduce per step).
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib

BLOCK_TENSORS = ["norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight",
                 "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
                 "mlp.fc2.weight", "mlp.fc2.bias"]
MAX_BLOCKS = 8


class S2Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim", "heads", "mlp_dim", "tokens", "tokens_pad", "n_blocks",
                                         "enable_pe")] + [("ln_eps", C.c_float)]


_P, _I, _F = C.c_void_p, C.c_int, C.c_float
_lib.register_signatures({
    "dvt_s2_param_offsets": (_I, [C.POINTER(S2Config), C.POINTER(C.c_int64)]),
    "dvt_s2_workspace_bytes": (C.c_int64, [C.POINTER(S2Config), _I, _I]),
    "dvt_s2_forward": (_I, [C.POINTER(S2Config), _P, _P, _P, _I, _P, C.c_int64, _P]),
    "dvt_s2_train_step": (_I, [C.POINTER(S2Config), _P, _P, _P, _P, _P, _I, _P, C.c_int64, _P, _P]),
    "dvt_adamw_step": (_I, [_P, _P, _P, _P, C.c_int64, _F, _F, _F, _F, _F, _I, _F, _P]),
})


def make_config(feat_dim: int, tokens: int, num_blocks: int = 1, enable_pe: bool = True) -> S2Config:
    if feat_dim not in (384, 768, 1024):
        raise NotImplementedError(f"stage-2 kernels are built for feat_dim 384 / 768 / 1024, got {feat_dim}")
    if not 1 <= num_blocks <= MAX_BLOCKS:
        raise ValueError(f"num_blocks must be in [1, {MAX_BLOCKS}]")
    c = S2Config()
    c.dim, c.heads, c.mlp_dim = feat_dim, feat_dim // 64, 4 * feat_dim
    c.tokens, c.tokens_pad = tokens, (tokens + 63) // 64 * 64
    c.n_blocks, c.enable_pe, c.ln_eps = num_blocks, int(bool(enable_pe)), 1e-6
    return c


def tensor_shapes(cfg: S2Config) -> list:
    d, f = cfg.dim, cfg.mlp_dim
    return [(d,), (d,), (3 * d, d), (3 * d,), (d, d), (d,), (d,), (d,), (f, d), (f,), (d, f), (d,)]


def param_layout(cfg: S2Config):
    """-> (total floats, {name: (offset, shape)}) with the reference's state-dict names
    (`pos_embed`, `denoiser.<tensor>` or `denoiser.<block>.<tensor>` for num_blocks > 1)."""
    out = (C.c_int64 * (2 + 12 * cfg.n_blocks))()
    _lib.check(_lib.lib().dvt_s2_param_offsets(C.byref(cfg), out), "dvt_s2_param_offsets")
    names = {}
    if cfg.enable_pe:
        names["pos_embed"] = (out[0], (1, cfg.tokens, cfg.dim))
    shapes = tensor_shapes(cfg)
    for b in range(cfg.n_blocks):
        prefix = "denoiser." if cfg.n_blocks == 1 else f"denoiser.{b}."
        for i, t in enumerate(BLOCK_TENSORS):
            names[prefix + t] = (out[1 + 12 * b + i], shapes[i])
    return int(out[1 + 12 * cfg.n_blocks]), names


class Stage2Engine:
    def __init__(self, cfg: S2Config, device: torch.device, inference_only: bool = False):
        """inference_only: a parameter arena and nothing else (no gradient / moment arenas: 3/4 of the memory) --
        what `Denoiser._engine_for` needs for its resized copies."""
        if torch.device(device).type != "cuda":
            raise _lib.DvtError("the stage-2 engine needs a HIP device; there is no CPU fallback")
        self.cfg, self.device = cfg, torch.device(device)
        self.total, self.layout = param_layout(cfg)
        z = lambda: torch.zeros(self.total, device=self.device, dtype=torch.float32)  # noqa: E731
        self.params = z()
        self.inference_only = inference_only
        self.grads = self.exp_avg = self.exp_avg_sq = None
        if not inference_only:
            self.grads, self.exp_avg, self.exp_avg_sq = z(), z(), z()
        self.loss = torch.zeros(4, device=self.device, dtype=torch.float32)
        self.step = 0
        self.param_version = 0  # bumped by everything that writes the parameter arena through this engine
        self._work = {}

    # ---- parameters -------------------------------------------------------------------------------
    def views(self, arena: torch.Tensor | None = None) -> dict:
        arena = self.params if arena is None else arena
        return {n: arena[o:o + math.prod(s)].view(s) for n, (o, s) in self.layout.items()}

    def init_parameters(self, generator: torch.Generator | None = None) -> None:
        """The reference's initial state: standalone timm `Block`s keep the `nn.Linear` / `nn.LayerNorm`
        defaults (kaiming-uniform(a=sqrt 5) weights and U(+-1/sqrt(fan_in)) biases; ones / zeros), and
        pos_embed = randn * 0.02 (online_denoiser.py:24-57).  Drawn on the host, then uploaded."""
        g = generator
        for name, v in self.views().items():
            if name == "pos_embed":
                t = torch.randn(v.shape, generator=g) * 0.02
            elif name.endswith("norm1.weight") or name.endswith("norm2.weight"):
                t = torch.ones(v.shape)
            elif name.endswith("norm1.bias") or name.endswith("norm2.bias"):
                t = torch.zeros(v.shape)
            else:
                wname = name[:-len("bias")] + "weight" if name.endswith("bias") else name
                fan_in = self.layout[wname][1][1]
                bound = 1.0 / math.sqrt(fan_in)
                t = (torch.rand(v.shape, generator=g) * 2 - 1) * bound
            v.copy_(t.to(self.device))

    def load_named(self, state: dict) -> None:
        v = self.views()
        missing = [k for k in v if k not in state]
        if missing:
            raise KeyError(f"stage-2 checkpoint lacks {missing}")
        for k, dst in v.items():
            dst.copy_(state[k].to(self.device, torch.float32).reshape(dst.shape))
        self.param_version += 1

    # ---- kernels ----------------------------------------------------------------------------------
    def _workspace(self, batch: int, training: bool) -> torch.Tensor:
        key = (batch, training)
        if key not in self._work:
            n = _lib.lib().dvt_s2_workspace_bytes(C.byref(self.cfg), batch, int(training))
            if n <= 0:
                raise _lib.DvtError("dvt_s2_workspace_bytes: invalid configuration")
            self._work = {k: v for k, v in self._work.items() if k[1] != training}  # one per mode
            self._work[key] = torch.empty(n, device=self.device, dtype=torch.uint8)
        return self._work[key]

    def _check_io(self, *ts):
        _lib.require_cuda(*ts)
        for t in ts:
            if t is None:
                continue
            if t.dtype != torch.float32 or not t.is_contiguous() or t.shape[1:] != (self.cfg.tokens, self.cfg.dim):
                raise _lib.DvtError(f"expected contiguous fp32 [batch, {self.cfg.tokens}, {self.cfg.dim}], got "
                                    f"{tuple(t.shape)} {t.dtype}")

    def forward(self, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        self._check_io(x, out)
        out = torch.empty_like(x) if out is None else out
        w = self._workspace(x.shape[0], False)
        _lib.check(_lib.lib().dvt_s2_forward(C.byref(self.cfg), _lib.ptr(self.params), _lib.ptr(x), _lib.ptr(out),
                                             x.shape[0], _lib.ptr(w), w.numel(), _lib.stream()), "dvt_s2_forward")
        return out

    def train_step(self, x: torch.Tensor, target: torch.Tensor, pred: torch.Tensor | None = None) -> torch.Tensor:
        """Gradients of this batch are ADDED to `self.grads`; returns the device tensor
        [loss, l2_loss, cosine_similarity_loss, 0] (no synchronisation)."""
        self._check_io(x, target, pred)
        if self.inference_only:
            raise _lib.DvtError("this engine was built inference_only (no gradient arena)")
        if target.shape != x.shape:
            raise _lib.DvtError("target and input shapes differ")
        w = self._workspace(x.shape[0], True)
        _lib.check(_lib.lib().dvt_s2_train_step(C.byref(self.cfg), _lib.ptr(self.params), _lib.ptr(self.grads),
                                                _lib.ptr(x), _lib.ptr(target), _lib.ptr(pred), x.shape[0], _lib.ptr(w),
                                                w.numel(), _lib.ptr(self.loss), _lib.stream()), "dvt_s2_train_step")
        return self.loss

    def adamw_step(self, lr: float, weight_decay: float, betas=(0.9, 0.999), eps: float = 1e-8,
                   grad_scale: float = 1.0) -> None:
        if self.inference_only:
            raise _lib.DvtError("this engine was built inference_only (no optimizer state)")
        self.step += 1
        self.param_version += 1
        _lib.check(_lib.lib().dvt_adamw_step(_lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.exp_avg),
                                             _lib.ptr(self.exp_avg_sq), self.total, lr, betas[0], betas[1], eps,
                                             weight_decay, self.step, grad_scale, _lib.stream()), "dvt_adamw_step")
