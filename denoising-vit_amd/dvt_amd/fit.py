"""FitEngine: the fused per-image fit of DVT stage 1 on one MI355X.

Host-side mirror of `denoise_an_image` (reference main_img_denoising.py:28-149): builds the
parameter arena (hash grid F, field MLP, shared artifacts G, residual predictor h), the Adam
state and the index stream, then hands the whole inner loop (:67-89) to native code
(`dvt_fit_run`, csrc/dvt_fit.hip) -- PyTorch only owns the memory.  Parameters can be
imported from / exported to the reference-style modules (`SingleImageDenoiser`,
`NeuralFeatureField`) so both APIs describe the same model.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib
from .utils.misc import lr_schedule


@dataclass
class FitSettings:
    """Flags of main_img_denoising.py:152-208 that shape the fit (same names, same defaults)."""
    feat_dim: int = 768
    noise_map_height: int = 37
    noise_map_width: int = 37
    n_levels: int = 16
    n_features_per_level: int = 8
    base_resolution: int = 16
    max_resolution: int = 1024
    log2_hashmap_size: int = 20
    num_iters: int = 25000
    warmup_iters: int = 2500
    freeze_shared_artifacts_after: float = 0.5
    lr: float = 0.01
    min_lr: float = 0.001
    weight_decay: float = 1e-5
    pixel_bsz: int = 2048
    enable_residual_predictor: bool = True
    grad_scale: float = 1024.0  # GradScaler(2**10) whose unscale_ is never called (:55, :88)
    beta1: float = 0.9
    beta2: float = 0.99
    eps: float = 1e-15
    grid_seed: int = 1337
    # "float32": fp32-operand MFMA in the MLP GEMMs.  "bfloat16": operands rounded to bf16 while
    # staged, fp32 accumulation and outputs -- the reference's `--dtype bfloat16` autocast mode
    # (main_img_denoising.py:78) with fp32 (instead of bf16) layer outputs.
    mlp_dtype: str = "float32"

    @property
    def lattice(self) -> int:
        return self.noise_map_height * self.noise_map_width


_TENSORS = [
    # (arena attr, shape fn)
    ("grid", lambda s, c: (int(c.grid.n_entries_total) * s.n_features_per_level,)),
    ("w1", lambda s, c: (c.hidden, s.n_levels * s.n_features_per_level)),
    ("b1", lambda s, c: (c.hidden,)),
    ("w2", lambda s, c: (s.feat_dim, c.hidden)),
    ("b2", lambda s, c: (s.feat_dim,)),
    ("G", lambda s, c: (s.lattice, s.feat_dim)),
    ("wh1", lambda s, c: (c.res_hidden, s.feat_dim)),
    ("bh1", lambda s, c: (c.res_hidden,)),
    ("wh2", lambda s, c: (c.res_hidden, c.res_hidden)),
    ("bh2", lambda s, c: (c.res_hidden,)),
    ("wh3", lambda s, c: (s.feat_dim, c.res_hidden)),
    ("bh3", lambda s, c: (s.feat_dim,)),
]


class FitEngine:
    def __init__(self, settings: FitSettings, n_rows: int, device: torch.device | str = "cuda"):
        self.s = settings
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.DvtError("FitEngine needs a HIP device; there is no CPU fallback")
        L = _lib.lib()
        cfg = _lib.FitConfig()
        cfg.feat_dim = settings.feat_dim
        cfg.hidden = settings.feat_dim // 2
        cfg.res_hidden = settings.feat_dim // 4
        cfg.lattice = settings.lattice
        cfg.n_rows = n_rows
        cfg.batch = settings.pixel_bsz
        cfg.num_iters = settings.num_iters
        cfg.switch_step = int(settings.freeze_shared_artifacts_after * settings.num_iters)
        cfg.enable_residual = int(settings.enable_residual_predictor)
        if settings.mlp_dtype not in ("float32", "bfloat16"):
            raise _lib.DvtError(f"mlp_dtype must be float32 or bfloat16, not {settings.mlp_dtype!r}")
        cfg.mlp_bf16 = int(settings.mlp_dtype == "bfloat16")
        cfg.grad_scale = settings.grad_scale
        cfg.beta1, cfg.beta2 = settings.beta1, settings.beta2
        cfg.eps, cfg.weight_decay = settings.eps, settings.weight_decay
        cfg.grid = _lib.grid_table(settings.n_levels, settings.n_features_per_level,
                                   settings.base_resolution, settings.max_resolution,
                                   settings.log2_hashmap_size)
        _lib.check(L.dvt_fit_layout(C.byref(cfg)), "dvt_fit_layout")
        self.cfg = cfg
        n = int(cfg.arena_floats)
        dev = self.device
        self.params = torch.zeros(n, device=dev, dtype=torch.float32)
        self.adam_m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.adam_v = torch.zeros(n, device=dev, dtype=torch.float32)
        self.grads = torch.zeros(n, device=dev, dtype=torch.float32)
        self.touched = torch.zeros(int(cfg.off_w1) // 256, device=dev, dtype=torch.int32)
        self.workspace = torch.empty(int(L.dvt_fit_workspace_floats(C.byref(cfg))), device=dev,
                                     dtype=torch.float32)
        self.losses = torch.zeros((settings.num_iters, 8), device=dev, dtype=torch.float32)
        self.h_lr = np.ascontiguousarray(
            [lr_schedule(i, settings.lr, settings.min_lr, settings.warmup_iters, settings.num_iters)
             for i in range(settings.num_iters)], dtype=np.float64)
        self._infer_ws = None

    # ------------------------------------------------------------------ parameter views
    def view(self, name: str) -> torch.Tensor:
        """A view of one tensor inside the arena (reference-shaped, except G = [H*W, C])."""
        off = int(getattr(self.cfg, "off_" + name))
        shape = dict(_TENSORS)[name](self.s, self.cfg)
        return self.params[off: off + math.prod(shape)].view(shape)

    def reset(self, generator: torch.Generator | None = None) -> None:
        """Fresh models for a new image, initialised like the reference constructors:
        grid U(-1e-4, 1e-4) with the SAME seed for every image (tcnn seed=1337), nn.Linear
        default init (kaiming_uniform(a=sqrt 5) -> U(+-1/sqrt(fan_in)) for weight and bias),
        G ~ N(0, 0.02^2) (offline_denoiser.py:33-36).  Adam state and gradients are cleared."""
        dev = self.device
        self.params.zero_()
        g = torch.Generator(device=dev).manual_seed(self.s.grid_seed)
        self.view("grid").uniform_(-1e-4, 1e-4, generator=g)
        for wname, bname in (("w1", "b1"), ("w2", "b2"), ("wh1", "bh1"), ("wh2", "bh2"),
                             ("wh3", "bh3")):
            w = self.view(wname)
            bound = 1.0 / math.sqrt(w.shape[1])
            w.uniform_(-bound, bound, generator=generator)
            self.view(bname).uniform_(-bound, bound, generator=generator)
        self.view("G").normal_(0.0, 0.02, generator=generator)
        self.adam_m.zero_()
        self.adam_v.zero_()
        self.grads.zero_()
        self.touched.zero_()

    def load_modules(self, denoiser, neural_field) -> None:
        """Copy the parameters of reference-style modules into the arena (and clear state)."""
        with torch.no_grad():
            self.params.zero_()
            self.view("grid").copy_(neural_field.neural_field.params.detach().to(self.device))
            self.view("w1").copy_(neural_field.mlp[0].weight.detach())
            self.view("b1").copy_(neural_field.mlp[0].bias.detach())
            self.view("w2").copy_(neural_field.mlp[2].weight.detach())
            self.view("b2").copy_(neural_field.mlp[2].bias.detach())
            G = denoiser.shared_artifacts.detach()  # [1, C, H, W] -> [H*W, C]
            self.view("G").copy_(G.permute(0, 2, 3, 1).reshape(self.s.lattice, self.s.feat_dim))
            if self.s.enable_residual_predictor:
                rp = denoiser.residual_predictor
                for i, (wn, bn) in zip((0, 2, 4), (("wh1", "bh1"), ("wh2", "bh2"), ("wh3", "bh3"))):
                    self.view(wn).copy_(rp[i].weight.detach())
                    self.view(bn).copy_(rp[i].bias.detach())
        self.adam_m.zero_()
        self.adam_v.zero_()
        self.grads.zero_()
        self.touched.zero_()

    def export_modules(self, denoiser, neural_field) -> None:
        with torch.no_grad():
            neural_field.neural_field.params.copy_(self.view("grid"))
            neural_field.mlp[0].weight.copy_(self.view("w1"))
            neural_field.mlp[0].bias.copy_(self.view("b1"))
            neural_field.mlp[2].weight.copy_(self.view("w2"))
            neural_field.mlp[2].bias.copy_(self.view("b2"))
            H, W, Cc = self.s.noise_map_height, self.s.noise_map_width, self.s.feat_dim
            denoiser.shared_artifacts.copy_(self.view("G").view(1, H, W, Cc).permute(0, 3, 1, 2))
            if self.s.enable_residual_predictor:
                rp = denoiser.residual_predictor
                for i, (wn, bn) in zip((0, 2, 4), (("wh1", "bh1"), ("wh2", "bh2"), ("wh3", "bh3"))):
                    rp[i].weight.copy_(self.view(wn))
                    rp[i].bias.copy_(self.view(bn))

    # ------------------------------------------------------------------ the loop
    @staticmethod
    def sample_indices(n_rows: int, num_iters: int, batch: int) -> np.ndarray:
        """The reference's index stream (main_img_denoising.py:73): `np.random.randint(0, N, B)`
        per step from the process-global MT19937 (seeded once by fix_random_seeds, quirk Q6).
        One [num_iters, B] draw consumes the stream exactly like num_iters successive draws."""
        return np.random.randint(0, n_rows, (num_iters, batch)).astype(np.int32)

    def buffers(self, feat: torch.Tensor, xy: torch.Tensor,
                idx: torch.Tensor | np.ndarray | None = None, log_every: int = 1000):
        """Validate the inputs, upload the index stream (on the current stream) and return the
        DvtFitBuffers descriptor of this fit."""
        s, cfg = self.s, self.cfg
        _lib.require_cuda(feat, xy)
        if feat.dtype != torch.float32 or xy.dtype != torch.float32:
            raise _lib.DvtError("feat/xy must be fp32")
        if feat.shape != (cfg.n_rows, s.feat_dim) or xy.shape != (cfg.n_rows, 2):
            raise _lib.DvtError(f"feat {tuple(feat.shape)} / xy {tuple(xy.shape)} do not match "
                                f"n_rows={cfg.n_rows}, C={s.feat_dim}")
        if not feat.is_contiguous() or not xy.is_contiguous():
            raise _lib.DvtError("feat/xy must be contiguous")
        # The reference asserts 0 <= coords <= 1 with a host sync on EVERY step
        # (neural_feature_field.py:47); here the whole coordinate table of the image is checked
        # once, asynchronously: the flag is read by check_inputs() after the fit (the retiring thread
        # of the driver, loss_log()), never inside the launch loop.
        lo, hi = torch.aminmax(xy)
        self._range_bad = (lo < 0) | (hi > 1) | torch.isnan(lo + hi)
        if idx is None:
            idx = self.sample_indices(cfg.n_rows, s.num_iters, s.pixel_bsz)
        if isinstance(idx, np.ndarray):
            # pinned staging buffer -> asynchronous H2D on the current stream (a pageable copy
            # would block the host until everything queued before it has drained)
            if getattr(self, "_idx_pin", None) is None:
                self._idx_pin = torch.empty((s.num_iters, s.pixel_bsz), dtype=torch.int32,
                                            pin_memory=True)
                self._idx_free = torch.cuda.Event()
                self._idx_free.record()
            self._idx_free.synchronize()  # the previous upload from this buffer has finished
            self._idx_pin.numpy()[...] = idx
            idx = self._idx_pin.to(self.device, non_blocking=True)
            self._idx_free.record()
        if idx.dtype != torch.int32 or tuple(idx.shape) != (s.num_iters, s.pixel_bsz):
            raise _lib.DvtError("idx must be int32 [num_iters, pixel_bsz]")
        self._idx = idx.contiguous()  # keep alive while kernels are in flight
        self._feat, self._xy = feat, xy
        b = _lib.FitBuffers()
        b.feat, b.xy, b.idx = feat.data_ptr(), xy.data_ptr(), self._idx.data_ptr()
        b.params, b.adam_m, b.adam_v = (self.params.data_ptr(), self.adam_m.data_ptr(),
                                        self.adam_v.data_ptr())
        b.grads, b.touched = self.grads.data_ptr(), self.touched.data_ptr()
        b.workspace, b.losses = self.workspace.data_ptr(), self.losses.data_ptr()
        b.h_lr = self.h_lr.ctypes.data
        b.log_every = int(log_every)
        return b

    def fit(self, feat: torch.Tensor, xy: torch.Tensor, idx: torch.Tensor | np.ndarray | None = None,
            log_every: int = 1000, step_begin: int = 0, step_end: int | None = None) -> None:
        """Enqueue Adam steps [step_begin, step_end) on the current stream (asynchronous).

        feat [n_rows, C] fp32 = all_raw_features.reshape(-1, C); xy [n_rows, 2] fp32 =
        all_pixel_coords.reshape(-1, 2); idx [num_iters, pixel_bsz] int32 row indices
        (default: the reference's numpy stream)."""
        b = self.buffers(feat, xy, idx, log_every)
        end = self.s.num_iters if step_end is None else step_end
        _lib.check(_lib.lib().dvt_fit_run(C.byref(self.cfg), C.byref(b), step_begin, end,
                                          _lib.stream()), "dvt_fit_run")

    @property
    def range_flag(self):
        """Device flag of the LAST `buffers()` call (a pipelined driver must keep it with the image: the
        engine is already preparing the next image when this one retires)."""
        return getattr(self, "_range_bad", None)

    _OWN_FLAG = object()

    def check_inputs(self, flag=_OWN_FLAG) -> None:
        """Raise if the coordinates handed to a fit left [0, 1] (synchronises on the flag).  Without an argument
        the flag of this engine's LAST fit is checked; the pipelined driver passes the flag that travelled with the
        image it retires -- `None` there means "this image was never fitted" and is an error, not a reason to look at
        the engine's current flag (which already belongs to the next image)."""
        if flag is FitEngine._OWN_FLAG:
            flag = self.range_flag
        elif flag is None:
            raise _lib.DvtError("check_inputs(None): the retired image carries no coordinate-range flag (never fitted?)")
        if flag is not None and bool(flag.item()):
            raise _lib.DvtError("coordinates should be in [0, 1] (neural_feature_field.py:47): the fit of this "
                                "image consumed out-of-range coordinates, its result is invalid")

    def infer(self, xy: torch.Tensor) -> torch.Tensor:
        """F(xy): the denoised features saved by the reference (quirk Q7) -- the field evaluated
        on a coordinate lattice, main_img_denoising.py:121-130 / offline_denoiser.py:151."""
        _lib.require_cuda(xy)
        shape = xy.shape[:-1]
        xy2 = xy.reshape(-1, 2).contiguous().float()
        n = xy2.shape[0]
        E = self.s.n_levels * self.s.n_features_per_level
        need = ((n * E + 255) // 256 * 256) + n * self.cfg.hidden + 256
        if self._infer_ws is None or self._infer_ws.numel() < need:
            self._infer_ws = torch.empty(need, device=self.device, dtype=torch.float32)
        out = torch.empty((n, self.s.feat_dim), device=self.device, dtype=torch.float32)
        _lib.check(_lib.lib().dvt_field_infer(C.byref(self.cfg), self.params.data_ptr(),
                                              xy2.data_ptr(), out.data_ptr(),
                                              self._infer_ws.data_ptr(), n, _lib.stream()),
                   "dvt_field_infer")
        return out.reshape(*shape, self.s.feat_dim)

    def loss_log(self) -> dict[int, dict[str, float]]:
        """Loss scalars of the logged steps (one D2H copy, after the loop)."""
        self.check_inputs()
        host = self.losses.cpu().numpy()
        keys = ("loss", "patch_l2_loss", "cosine_similarity_loss", "residual_loss",
                "residual_sparsity_loss")
        return {i: dict(zip(keys, map(float, host[i, :5]))) for i in range(host.shape[0])
                if host[i, 0] != 0.0}


FIT_BATCH_MAX = 4    # DVT_FIT_BATCH_MAX: fits that share every launch of a step
FIT_CONCURRENT_MAX = 16  # fits advanced concurrently by one fit_many call (groups of FIT_BATCH_MAX on side streams)

_side_streams: dict = {}


def _fit_group(engines, feats, xys, idxs, log_every, step_begin, step_end) -> None:
    k = len(engines)
    bufs = [e.buffers(f, x, i, log_every) for e, f, x, i in zip(engines, feats, xys, idxs)]
    buf_arr = (C.POINTER(_lib.FitBuffers) * k)(*[C.pointer(b) for b in bufs])
    end = engines[0].s.num_iters if step_end is None else step_end
    _lib.check(_lib.lib().dvt_fit_run_batched(C.byref(engines[0].cfg), k, buf_arr, step_begin, end,
                                              _lib.stream()), "dvt_fit_run_batched")


def fit_many(engines, feats, xys, idxs=None, log_every: int = 1000, step_begin: int = 0,
             step_end: int | None = None) -> None:
    """Advance k independent fits (k images, identical settings) concurrently -- BASELINE configs[2], "many concurrent
    neural fields per GPU".  Up to FIT_BATCH_MAX fits share every launch of a step (`dvt_fit_run_batched`, blockIdx.y =
    fit); more than that (k <= FIT_CONCURRENT_MAX) run as groups of FIT_BATCH_MAX on side streams, one host thread per
    group (the native call is back-pressured by the HIP queue for the length of the fit), joined back into the CURRENT
    stream.  Measured (tools/bench_fit_batch.py, profiles/r03): one fit alone is latency-bound (94 us per step at
    C = 768), two or more concurrent fits saturate the GPU at 68-72 us per step and fit whether they share launches or
    streams -- 1.4 x, flat from K = 4 to 16.  Each engine keeps its own arena / Adam state / index stream; index streams
    are drawn in engine order from the reference's numpy RNG when `idxs` is None (before any group starts)."""
    k = len(engines)
    if not 1 <= k <= FIT_CONCURRENT_MAX:
        raise _lib.DvtError(f"fit_many takes 1..{FIT_CONCURRENT_MAX} engines")
    ref = bytes(engines[0].cfg)
    if any(bytes(e.cfg) != ref for e in engines[1:]):
        raise _lib.DvtError("batched fits must share one configuration")
    if len({id(e) for e in engines}) != k:
        raise _lib.DvtError("batched fits need distinct engines")
    s0 = engines[0].s  # missing index streams: the reference's draw order, image by image, before any group starts
    idxs = [i if i is not None else FitEngine.sample_indices(engines[0].cfg.n_rows, s0.num_iters, s0.pixel_bsz)
            for i in (idxs if idxs is not None else [None] * k)]
    if k <= FIT_BATCH_MAX:
        _fit_group(engines, feats, xys, idxs, log_every, step_begin, step_end)
        return
    import threading
    dev = torch.device(engines[0].device)
    if dev.index is None:  # worker threads and stream objects need the concrete device
        dev = torch.device("cuda", torch.cuda.current_device())
    cur = torch.cuda.current_stream(dev)
    groups = [range(i, min(k, i + FIT_BATCH_MAX)) for i in range(0, k, FIT_BATCH_MAX)]
    # side streams inherit the calling stream's priority: a group beyond the first must not lose the fit's standing
    # against the extractor stream
    prio = int(getattr(cur, "priority", 0))
    pool = _side_streams.setdefault((torch.device(dev), prio), [])
    while len(pool) < len(groups):
        pool.append(torch.cuda.Stream(device=dev, priority=prio))
    errors = []

    def work(gi, members):
        try:
            torch.cuda.set_device(dev)
            side = pool[gi]
            side.wait_stream(cur)  # inputs produced on the caller's stream
            with torch.cuda.stream(side):
                _fit_group([engines[j] for j in members], [feats[j] for j in members], [xys[j] for j in members],
                           [idxs[j] for j in members], log_every, step_begin, step_end)
        except Exception as exc:  # surfaced on the calling thread
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(gi, m), daemon=True) for gi, m in enumerate(groups)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for gi in range(len(groups)):
        cur.wait_stream(pool[gi])
    if errors:
        raise errors[0]
