"""Stage-1 driver: per image, frozen-ViT features of 768 random views + the original, then
the fused neural-field fit, then the two .npy outputs.

Mirrors the reference's main_img_denoising.py (CLI flags :152-208 with the same names and
defaults, work-list handling :226-234, per-image flow :301-352, output layout :131-146,
resume by file existence :303-307) and the sharding of sample_scripts/stage1.sh (one
process per GPU, disjoint contiguous slices, no collective in the loop).

    python -m dvt_amd.stage1 --img_path demo/cat.jpg --num_iters 1000 --warmup_iters 100 \\
        --data_root demo --save_root out/

Multi-GPU: launch with torch.distributed.run (one rank per GPU); each rank takes
`misc.shard_range(start_idx, num_imgs, rank, world)` of the work-list.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import queue
import threading
import time

import numpy as np
import torch

from . import _lib
from . import dist as D
from . import views as V
from ._lib import DvtError
from .fit import FIT_CONCURRENT_MAX, FitEngine, FitSettings, fit_many
from .models import MODEL_LIST, PretrainedViTWrapper
from .utils import misc



class _ExplicitInt(argparse.Action):
    """int option that also records that it was GIVEN (namespace.<dest>_explicit): --extract_bsz keeps the reference's default
    of 32 in the table, but only a value the user typed bounds the extractor's launches."""

    def __call__(self, parser, namespace, values, option_string=None):
        setattr(namespace, self.dest, int(values))
        setattr(namespace, self.dest + "_explicit", True)


def get_args(argv=None):
    p = argparse.ArgumentParser(description="DVT Stage-1: Single Image Denoising (MI355X)")
    p.add_argument("--model", type=str, default="vit_base_patch14_dinov2.lvd142m", choices=MODEL_LIST)
    p.add_argument("--input_size", type=int, default=518, nargs="+")
    p.add_argument("--stride_size", type=int, default=14)
    p.add_argument("--layer_depth_ratio", type=float, default=1.0)
    p.add_argument("--img_path", type=str, default="demo/assets/demo/cat.jpg")
    p.add_argument("--dtype", type=str, default="float32", choices=["float32", "bfloat16"],
                   help="float32 (reference default): fp32 extractor + fp32-operand fit, exact-fp32 matrix cores, "
                        "seconds per image; bfloat16: the reference's autocast mode, bf16 MFMA, ~10x faster")
    p.add_argument("--fp32_matmul", type=str, default="highest", choices=["highest", "high"],
                   help="with --dtype float32 only; torch.set_float32_matmul_precision's vocabulary.  highest (default, "
                        "what the reference runs with): exact-fp32 matrix cores.  high: the extractor's matrix products "
                        "(linear layers, q.k^T, p.v) as bf16x3 on the bf16 pipe (~1e-5 relative per product, fp32 "
                        "accumulation, 3x faster extractor); LayerNorm, softmax, GELU, the residual stream and the whole fit "
                        "stay fp32")
    p.add_argument("--data_root", type=str, default=None)
    p.add_argument("--save_root", type=str, default=None)
    p.add_argument("--start_idx", type=int, default=0)
    p.add_argument("--num_imgs", type=int, default=100)
    p.add_argument("--num_views", type=int, default=768)
    p.add_argument("--num_iters", type=int, default=25000)
    p.add_argument("--warmup_iters", type=int, default=2500)
    p.add_argument("--n_levels", type=int, default=16)
    p.add_argument("--freeze_shared_artifacts_after", type=float, default=0.5)
    p.add_argument("--lr", type=float, default=0.01)
    p.add_argument("--min_lr", type=float, default=0.001)
    p.add_argument("--weight_decay", type=float, default=1e-5)
    p.add_argument("--extract_bsz", type=int, default=32, action=_ExplicitInt,
                   help="the reference's DataLoader batch for feature extraction (main_img_denoising.py:196; its default 32). "
                        "Here the views are already on the device and results do not depend on the batching (tested), so the "
                        "number of views per extractor LAUNCH is its own knob, --extract_launch_views.  An EXPLICIT "
                        "--extract_bsz still bounds the launches (and with them the extractor's workspace, ~26 MB per view "
                        "for ViT-B/14) when --extract_launch_views is left at 0")
    p.add_argument("--pixel_bsz", type=int, default=2048)
    p.add_argument("--output_dir", type=str, default="./work_dirs/demo")
    p.add_argument("--num_vis_samples", type=int, default=5)
    p.add_argument("--vis_freq", type=int, default=100)
    p.add_argument("--seed", type=int, default=0)
    # additions of this build
    p.add_argument("--extract_launch_views", type=int, default=0,
                   help="cap on the views per extractor launch; 0 (default) = 400: 769 views -> 398 + 371 (two launches; "
                        "profiles/r04/r04g_*: 3 %% faster than 7 x 110, the tails of the GEMMs' tile rounds weigh less).  "
                        "The split below the cap is tile-round aware (dvt_amd.vit.plan_launches)")
    p.add_argument("--vit_checkpoint", type=str, default=None, help="timm-layout state dict (.pth)")
    p.add_argument("--synthetic", action="store_true", help="N(0,1) views instead of image crops")
    p.add_argument("--allow_random_vit", action="store_true",
                   help="run without a checkpoint on RANDOM ViT weights (tests / plumbing runs only; implied by "
                        "--synthetic).  Without it a missing checkpoint is an error, as in the reference.")
    p.add_argument("--fit_batch", type=int, default=0,
                   help="images fitted concurrently (BASELINE configs[2]): groups of 4 share every launch "
                        "(dvt_fit_run_batched), further groups run on side streams; 1..16.  0 (default) = auto: 4 (round 6: +0.9 .. "
                        "1.8 % images/s at 1000 iterations; 0.58 instead of 0.46 images/s at 20000) -- each image in flight "
                        "holds its views + feature store (5.7 GB at ViT-B/14, 769 views); 1 = the reference's one fit at a time")
    args = p.parse_args(argv)
    if isinstance(args.input_size, int):
        args.input_size = (args.input_size, args.input_size)
    elif len(args.input_size) == 1:
        args.input_size = (args.input_size[0], args.input_size[0])
    args.input_size = tuple(args.input_size)
    assert args.input_size[0] % args.stride_size == 0, "height must be divisible by stride_size"
    assert args.input_size[1] % args.stride_size == 0, "width must be divisible by stride_size"
    return args


def work_list(args) -> list[str]:
    """main_img_denoising.py:226-234."""
    if os.path.isfile(args.img_path):
        if args.img_path.endswith("txt"):
            with open(args.img_path) as f:
                names = f.read().splitlines()
        else:
            names = [args.img_path]
    else:
        names = glob.glob(os.path.join(args.img_path, "**/*"), recursive=True)
    return names[args.start_idx: args.start_idx + args.num_imgs]


def _cu_masked_stream(device, keep_per_32: int, first: int = 0):
    """A HIP stream whose kernels may only run on `keep_per_32` of every 32 CUs
    (hipExtStreamCreateWithCUMask), wrapped for torch.  Leaves a few CUs permanently free of the
    extractor's long-running workgroups so that the fit's short dependent launches start at once."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    n_cu = torch.cuda.get_device_properties(device).multi_processor_count
    words = (n_cu + 31) // 32
    mask = (ctypes.c_uint32 * words)(*([(((1 << keep_per_32) - 1) << first) & 0xFFFFFFFF] * words))
    stream = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(stream), ctypes.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(stream.value, device=device)


class _Slot:
    """One image in flight: its views, coordinates, feature store and host output buffers."""

    def __init__(self, n, size, pos_h, pos_w, feat_dim, dev):
        self.views = torch.zeros((n, 3, *size), device=dev)
        self.coords = torch.zeros((n, pos_h, pos_w, 2), device=dev)
        self.features = torch.zeros((n, pos_h, pos_w, feat_dim), device=dev)
        self.raw_host = torch.empty((pos_h, pos_w, feat_dim), pin_memory=True)
        self.den_host = torch.empty((1, pos_h, pos_w, feat_dim), pin_memory=True)
        self.extracted = torch.cuda.Event()
        self.fitted = torch.cuda.Event()
        self.tag = None
        self.range_flag = None


class Stage1:
    """Everything that is reused across images on one GPU: ViT weights, the view / feature /
    coordinate buffers (main_img_denoising.py:261-276) and the fit engine.

    Images are pipelined over two HIP streams in groups of `fit_batch`: the MFMA-bound extractor
    of the next group runs on `s_vit` while the latency/HBM-bound fits of the current group run
    -- batched into shared launches (`fit_many`) -- on the high-priority `s_fit` (`depth` groups
    of buffers; depth=1, fit_batch=1 reproduces the reference's strictly serial flow)."""

    def __init__(self, args, device, vit: PretrainedViTWrapper | None = None, depth: int = 2,
                 vit_cus_per_32: int = 32, fit_batch: int = 1):
        # argument checks first: nothing is allocated for a run that cannot start
        self.extract_matmul = str(getattr(args, "fp32_matmul", "highest") or "highest")
        if self.extract_matmul not in ("highest", "high"):
            raise DvtError(f"--fp32_matmul must be highest or high, not {self.extract_matmul!r}")
        self.args, self.device = args, torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:  # worker threads call set_device
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.vit = vit or PretrainedViTWrapper(
            args.model, stride=args.stride_size, checkpoint_path=getattr(args, "vit_checkpoint", None),
            img_size=args.input_size, dtype=str(getattr(args, "dtype", "float32")),
            allow_random_init=bool(getattr(args, "synthetic", False) or getattr(args, "allow_random_vit", False)))
        v = self.vit
        self.layer_index = int(args.layer_depth_ratio * v.last_layer_index)
        self.pos_h = (args.input_size[0] - v.patch_size) // args.stride_size + 1
        self.pos_w = (args.input_size[1] - v.patch_size) // args.stride_size + 1
        self.feat_dim = v.n_output_dims
        n = args.num_views + 1
        dev = self.device
        depth = max(1, depth)
        self.fit_batch = kb = min(max(1, fit_batch), FIT_CONCURRENT_MAX) if depth > 1 else 1
        self.depth = depth
        self.slots = [_Slot(n, args.input_size, self.pos_h, self.pos_w, self.feat_dim, dev)
                      for _ in range(depth * kb)]
        s = FitSettings(feat_dim=self.feat_dim, noise_map_height=self.pos_h,
                        noise_map_width=self.pos_w, n_levels=args.n_levels,
                        num_iters=args.num_iters, warmup_iters=args.warmup_iters,
                        freeze_shared_artifacts_after=args.freeze_shared_artifacts_after,
                        lr=args.lr, min_lr=args.min_lr, weight_decay=args.weight_decay,
                        pixel_bsz=args.pixel_bsz,
                        mlp_dtype="bfloat16" if str(getattr(args, "dtype", "float32")) in
                        ("bfloat16", "bf16", "torch.bfloat16") else "float32")
        self.engines = [FitEngine(s, n * self.pos_h * self.pos_w, dev) for _ in range(kb)]
        self.engine = self.engines[0]
        self.gen = torch.Generator(device=dev).manual_seed(args.seed)
        if depth > 1:
            # Neither stream is favoured by the queue arbiter.  Measured (profiles/r02/r02h_*): fit favoured 2.28 images/s
            # (extractor GEMMs 866 us), extractor favoured 2.15 (GEMMs 629 us, but the fit starves), neither 2.31.
            prio = os.environ.get("DVT_STREAM_PRIO", "none")
            self.s_vit = (torch.cuda.Stream(device=dev, priority=-1 if prio == "vit" else 0) if vit_cus_per_32 >= 32
                          else _cu_masked_stream(dev, vit_cus_per_32))
            fit_cus = int(os.environ.get("DVT_FIT_CUS", "32"))  # experiment: confine the fit to the LAST n of every 32 CUs
            self.s_fit = (torch.cuda.Stream(device=dev, priority=-1 if prio == "fit" else 0) if fit_cus >= 32
                          else _cu_masked_stream(dev, fit_cus, 32 - fit_cus))
        else:
            self.s_vit = self.s_fit = torch.cuda.current_stream(dev)
        # views per extractor launch: --extract_launch_views, else an EXPLICIT --extract_bsz (the reference's DataLoader
        # batch; a user who passes a small one to bound memory keeps that bound -- ADVICE r4), else 400
        elv = int(getattr(args, "extract_launch_views", 0) or 0)
        ebs = getattr(args, "extract_bsz", None) if getattr(args, "extract_bsz_explicit", False) else None
        self.extract_launch_views = max(1, elv if elv > 0 else (min(400, int(ebs)) if ebs else 400))
        self._plan_logged = False
        # `--dtype` is the reference's one precision switch (main_img_denoising.py:173, :257): float32 = fp32
        # extractor AND fp32-operand fit (autocast off, its default); bfloat16 = both under bf16 autocast
        self.extract_dtype = ("bfloat16" if str(getattr(args, "dtype", "float32")) in
                              ("bfloat16", "bf16", "torch.bfloat16") else "float32")
        if self.extract_dtype != "float32":
            self.extract_matmul = "highest"  # the switch only exists for fp32 operands
        # Everything above was allocated / zero-filled on the CURRENT stream; the first writers are the
        # side streams.  One device-wide sync here orders them for good (a zero-fill must never land
        # after set_views / reset).
        torch.cuda.synchronize(dev)
        self.timings = []
        self._idx_spare = []  # index streams drawn ahead for the next fit (numpy stream order kept)
        self._idx_queue = None  # look-ahead queue while `run` is active

    def vit_launch_views(self, n_views: int) -> list:
        """Views of each extractor launch for an image of `n_views` views (what `extract` will do; reported by bench.py)."""
        eng = self.vit._engine(self.device, self.extract_dtype, self.extract_matmul)
        return eng.launch_plan(n_views, self.extract_launch_views)

    # -- single-image pieces (each enqueues on the CURRENT stream) -------------------------
    def extract(self, slot: _Slot) -> None:
        """Feature extraction of all views into the feature store (:315-339), NHWC, no NCHW round trip, in launches of at
        most `extract_launch_views` views (default cap 400: 769 views -> 398 + 371, dvt_amd.vit.plan_launches)."""
        if not self._plan_logged:  # once: what the launches and their workspace will be
            self._plan_logged = True
            eng = self.vit._engine(self.device, self.extract_dtype, self.extract_matmul)
            plan = eng.launch_plan(slot.views.shape[0], self.extract_launch_views)
            print(f"extractor: {slot.views.shape[0]} views per image in launches of {plan} views "
                  f"(cap {self.extract_launch_views}), workspace {eng.workspace_bytes(max(plan)) / 2**30:.1f} GiB, {self.extract_dtype}",
                  flush=True)
        with torch.no_grad():
            self.vit.features_nhwc(slot.views, self.layer_index, out=slot.features,
                                   max_batch=self.extract_launch_views, dtype=self.extract_dtype,
                                   matmul=self.extract_matmul)

    def fit(self, slot: _Slot, log_every: int = 1000) -> torch.Tensor:
        """denoise_an_image (:28-149): fresh models, the loop, then F on the original image's
        lattice (quirk Q7).  Returns denoised_feats [1, H, W, C] (device)."""
        return self.fit_group([slot], log_every)[0]

    def _draw_indices(self, k: int):
        """k index streams from the reference's numpy stream (main_img_denoising.py:73), in image order
        (2 M draws = ~20 ms of host time each)."""
        e = self.engine
        return [e.sample_indices(e.cfg.n_rows, e.s.num_iters, e.s.pixel_bsz) for _ in range(k)]

    def _next_indices(self, k: int):
        """Index streams of the next k images: from the look-ahead thread of `run` when it is active
        (drawn while the fit thread is blocked inside the previous image's launch loop), else drawn
        here.  Either way the numpy stream is consumed strictly in image order."""
        out = []
        while len(out) < k and self._idx_spare:
            out.append(self._idx_spare.pop(0))
        while len(out) < k:
            q = self._idx_queue
            out.append(q.get() if q is not None else self._draw_indices(1)[0])
        return out

    def fit_group(self, group, log_every: int = 1000):
        """The fits of up to `fit_batch` images advanced together (shared launches).  Models are
        constructed image by image in order (torch RNG), index streams are drawn in the same
        order (numpy RNG) -- the draws of the reference's sequential loop."""
        engines = self.engines[:len(group)]
        C = self.feat_dim
        auto_fused = engines[0].s.mlp_dtype == "float32" and self.depth > 1 and 6 not in _lib.user_tune
        if auto_fused:
            # fp32-operand fit beside a running extractor: the fused row kernel (round 5: 3 launches per step, 157 KB of LDS
            # per workgroup) against the layer-by-layer launches (10 per step, 17 KB each) -- alone they take the same 200 us
            # per step.  Beside the bf16 extractor (one 136-KB GEMM workgroup per CU) either waits for whole CUs and fewer
            # launches win (+2.5 % images/s); beside the fp32 extractor (two 64-KB GEMM workgroups per CU) the small kernels
            # slip in and the fused one costs 2.6 % (profiles/r05/r05f_*): the driver picks per mode.
            _lib.check(_lib.lib().dvt_tune_set(6, 2 if self.extract_dtype == "float32" else 3), "dvt_tune_set(6)")
        idxs = self._next_indices(len(group))
        for e in engines:
            e.reset(self.gen)
        try:
            fit_many(engines, [sl.features.view(-1, C) for sl in group],
                     [sl.coords.view(-1, 2) for sl in group], idxs, log_every=log_every)
        finally:
            # (the launches are enqueued and the library latches the choice per dvt_fit_run_batched call: back to the library's
            # default for whoever comes next -- only where THIS method changed it; a user's `--tune 6=.` is never touched)
            if auto_fused:
                _lib.lib().dvt_tune_set(6, 3)
        for e, sl in zip(engines, group):
            sl.range_flag = e.range_flag  # travels with the image; the engine moves on to the next one
        return [e.infer(sl.coords[-1]).unsqueeze(0) for e, sl in zip(engines, group)]

    # -- the pipeline --------------------------------------------------------------------------
    def taper_pays(self) -> bool:
        """Does a tapered tail (group_plan) pay?  Only where the EXTRACTOR sets the pace: a group's fits must be done before
        the next group is extracted.  With long fits (the reference's default 25 000 iterations: 5 s of fp32 fit per image
        against 1.9 s of extraction) the run is fit-bound, sharing every launch between 4 fits is what counts, and a tail of
        2 + 1 + 1 fits costs 5 s of a 22-s run (measured, profiles/r06/literal_defaults/).  A model, not a measurement: fit
        ~ 70 (bf16 operands) / 200 (fp32) us per step and fit at 4 fits per launch; extraction ~ 0.35 / 2.5 / 1.2 ms per view of a
        ViT-B/14 at 518 x 518 (bf16 / fp32 / fp32 "high"), scaled by depth x dim^2 x tokens.  DVT_FIT_TAPER=1 / 0 forces it."""
        env = os.environ.get("DVT_FIT_TAPER", "")
        if env in ("0", "1"):
            return env == "1"
        return self.taper_model(self.args.num_iters, self.args.num_views + 1, self.extract_dtype, self.extract_matmul,
                                int(self.layer_index) + 1, self.feat_dim, self.pos_h * self.pos_w + 1)

    @staticmethod
    def taper_model(num_iters: int, n_views: int, dtype: str, matmul: str, blocks: int, feat_dim: int, tokens: int) -> bool:
        fit_s = float(num_iters) * (70e-6 if dtype == "bfloat16" else 200e-6)
        per_view = 0.35e-3 if dtype == "bfloat16" else (1.2e-3 if matmul == "high" else 2.5e-3)
        scale = (max(1, blocks) / 12.0) * (feat_dim / 768.0) ** 2 * (tokens / 1370.0)
        return fit_s < n_views * per_view * scale

    @staticmethod
    def group_plan(total: int | None, kb: int, taper: bool = True) -> list | None:
        """Sizes of the fit groups of a run of `total` images (None: unknown -- greedy groups of `kb`).  Groups of `kb` share
        every fit launch (68-72 instead of 92 us per fit-step), but a group's fits start only when its LAST image is
        extracted: the run's last group would fit with nothing left to overlap -- 4 fits = 280 ms of a 20-image run.  So
        the tail tapers: ..., kb, kb, 2, 1, 1 (kb >= 4; 1, 1 for kb 2-3): the last extractions run beside the fits before
        them and only ONE fit (92 ms) drains alone.  A partial group, if any, goes first.  `taper` False (fit-bound runs,
        taper_pays): greedy groups."""
        if total is None or kb <= 1 or not taper:
            return None
        tail = [2, 1, 1] if kb >= 4 else [1, 1]
        if total <= sum(tail):
            return [1] * total
        body = total - sum(tail)
        return ([body % kb] if body % kb else []) + [kb] * (body // kb) + tail

    def run(self, jobs, on_result=None, log_every: int = 1000, total: int | None = None) -> int:
        """jobs: iterable of (tag, set_views) with set_views(slot) filling slot.views / slot.coords
        (called with `s_vit` current).  on_result(tag, raw_host, den_host) is called (on the
        retiring thread) once an image's outputs have landed in pinned memory.  Returns the number
        of images.  `total`: the number of jobs when the caller knows it (len(jobs) is tried) -- lets the last fit groups
        taper (group_plan); the results do not depend on the grouping.

        Host threads: the EXTRACTOR walks `jobs` and enqueues view synthesis + ViT on `s_vit`; the
        calling thread enqueues the fits on `s_fit`; the RETIRER waits for finished images, hands
        them to `on_result` (the .npy writer) and recycles their buffers; the INDEX thread draws
        the next images' index streams.  With one thread the ~10 k launches of a fit back-pressure
        the host for most of the fit's duration (the HIP queue holds ~650 launches) and both
        streams idled 25-85 ms per image waiting for it."""
        kb, dev = self.fit_batch, self.device
        if total is None and hasattr(jobs, "__len__"):
            total = len(jobs)
        plan = self.group_plan(total, kb, self.taper_pays())
        cur = torch.cuda.current_stream(dev)
        for side in (self.s_vit, self.s_fit):  # work queued by the caller (e.g. resident inputs) comes first
            if side != cur:
                side.wait_stream(cur)
        free, ready, fitted = queue.Queue(), queue.Queue(), queue.Queue()
        for slot in self.slots:
            free.put(slot)
        errors = []
        done = [0]

        def extractor():
            try:
                torch.cuda.set_device(dev)
                for tag, set_views in jobs:
                    slot = free.get()
                    if slot is None:  # another thread failed
                        return
                    slot.tag = tag
                    with torch.cuda.stream(self.s_vit):
                        set_views(slot)
                        self.extract(slot)
                        slot.extracted.record(self.s_vit)
                    ready.put(slot)
            except BaseException as e:  # noqa: BLE001 - re-raised on the calling thread
                errors.append(e)
            finally:
                ready.put(None)

        def retirer():
            try:
                torch.cuda.set_device(dev)
                while True:
                    group = fitted.get()
                    if group is None:
                        return
                    group[-1].fitted.synchronize()  # recorded after the whole group's D2H copies
                    for slot in group:
                        self.engine.check_inputs(slot.range_flag)  # asynchronous range check, already complete
                    for slot in group:
                        if on_result is not None:
                            on_result(slot.tag, slot.raw_host.numpy(), slot.den_host.numpy())
                        done[0] += 1
                        free.put(slot)
            except BaseException as e:  # noqa: BLE001
                errors.append(e)
                free.put(None)

        idx_q = queue.Queue(maxsize=2 * kb)
        stop_idx = threading.Event()
        undelivered = []

        def indexer():  # the only consumer of the global numpy stream while the pipeline runs
            try:
                while not stop_idx.is_set():
                    item = self._draw_indices(1)[0]
                    while not stop_idx.is_set():
                        try:
                            idx_q.put(item, timeout=0.05)
                            item = None
                            break
                        except queue.Full:
                            pass
                    if item is not None:  # stopped with one stream drawn but not handed over
                        undelivered.append(item)
            except BaseException as e:  # noqa: BLE001
                errors.append(e)

        self._idx_queue = idx_q
        threads = [threading.Thread(target=extractor, name="dvt-extractor", daemon=True),
                   threading.Thread(target=retirer, name="dvt-retirer", daemon=True),
                   threading.Thread(target=indexer, name="dvt-indices", daemon=True)]
        for th in threads:
            th.start()
        try:
            last = False
            gi = 0
            while not last and not errors:
                group = []
                want = plan[gi] if plan is not None and gi < len(plan) else kb
                gi += 1
                while len(group) < want:
                    slot = ready.get()
                    if slot is None:
                        last = True
                        break
                    group.append(slot)
                if not group:
                    break
                with torch.cuda.stream(self.s_fit):
                    for slot in group:
                        self.s_fit.wait_event(slot.extracted)
                    dens = self.fit_group(group, log_every)
                    for slot, den in zip(group, dens):
                        slot.raw_host.copy_(slot.features[-1], non_blocking=True)
                        slot.den_host.copy_(den, non_blocking=True)
                    group[-1].fitted.record(self.s_fit)
                fitted.put(group)
        except BaseException:
            free.put(None)  # unblock the extractor thread
            raise
        finally:
            fitted.put(None)
            stop_idx.set()
            for th in threads:
                th.join()
            self._idx_queue = None
            while True:  # streams drawn ahead but not used: first in line for the next run / fit
                try:
                    self._idx_spare.append(idx_q.get_nowait())
                except queue.Empty:
                    break
            self._idx_spare.extend(undelivered)  # drawn last
        if errors:
            raise errors[0]
        return done[0]

    def process(self, set_views, save_paths=None):
        """Strictly serial single image (reference flow) with the two timers of :341, :355."""
        slot = self.slots[0]
        set_views(slot)
        torch.cuda.synchronize(self.device)
        t0 = time.time()
        self.extract(slot)
        torch.cuda.synchronize(self.device)
        t1 = time.time()
        den = self.fit(slot)
        raw_h = slot.features[-1].float().cpu().numpy()
        den_h = den.float().cpu().numpy()
        t2 = time.time()
        self.timings.append({"t_extract": t1 - t0, "t_fit": t2 - t1})
        if save_paths is not None:
            misc.atomic_save_npy(save_paths[0], raw_h)  # [H, W, C]
            misc.atomic_save_npy(save_paths[1], den_h)  # [1, H, W, C]
        return raw_h, den_h


def main(args, rank: int = 0, world: int = 1, stage_factory=None, device=None):
    """The sweep of one rank.  `stage_factory(args, device)` builds the per-GPU engine (default:
    `Stage1`); tests inject a host-only stand-in to drive this function under gloo."""
    os.makedirs(args.output_dir, exist_ok=True)
    misc.fix_random_seeds(args.seed)
    if rank == 0:
        print(f"Arguments:\n{json.dumps(vars(args), indent=4, default=str)}")
    if device is None:
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    if device.type == "cuda":
        torch.cuda.set_device(device)
    dist_on = D.init(device, world)
    names = work_list(args)
    lo, hi = misc.shard_range(0, len(names), rank, world)
    names = names[lo:hi]
    if stage_factory is None:
        fb = int(getattr(args, "fit_batch", 0) or 0)
        if fb <= 0:  # auto (round 6): shared fit launches pay at every schedule measured, most where the fit bounds an image
            fb = 4
        st = Stage1(args, device, fit_batch=fb)
    else:
        st = stage_factory(args, device)
    norm = st.vit.transformation.transforms[-1]
    start = time.time()
    # Crop parameters come from their own generator: in the reference they are drawn by the
    # DataLoader worker processes (own seeds), not from the main process' numpy stream, which
    # serves only the fit's index draws (main_img_denoising.py:73) -- and here the two are
    # consumed by different host threads.
    view_rng = np.random.RandomState(args.seed + 1000003 * (rank + 1))

    def jobs():
        for idx, filename in enumerate(names):
            filename = filename.strip().split(" ")[0]
            paths = None
            if args.data_root is not None:
                filename = os.path.join(args.data_root, filename)
                if misc.check_if_file_exists(args, filename):
                    print(f"Skipping {filename}")
                    continue
                paths = misc.output_paths(args.save_root, args.model, args.data_root, filename)

            def set_views(slot, filename=filename, idx=idx):
                if args.synthetic:
                    v, c = V.synthetic_views(args.num_views, args.input_size, st.pos_h, st.pos_w,
                                             device, seed=args.seed + idx)
                    slot.views.copy_(v)
                    slot.coords.copy_(c)
                else:
                    img = V.load_image(filename, args.input_size, norm.mean, norm.std, device)
                    boxes, coords = V.sample_view_boxes(args.num_views, args.input_size, st.pos_h,
                                                        st.pos_w, rng=view_rng)
                    V.render_views(img, boxes, slot.views)
                    slot.coords.copy_(coords.to(device), non_blocking=True)

            yield (filename, paths), set_views

    def on_result(tag, raw_h, den_h):
        filename, paths = tag
        if paths is not None:
            misc.atomic_save_npy(paths[0], raw_h.copy())  # [H, W, C]
            misc.atomic_save_npy(paths[1], den_h.copy())  # [1, H, W, C]
        el = time.time() - start
        print(f"[rank {rank}] {filename}: done at {el:.2f}s")
        with open(os.path.join(args.output_dir, f"timings_rank{rank}.jsonl"), "a") as f:
            f.write(json.dumps({"file": filename, "elapsed_s": el}) + "\n")

    done = st.run(jobs(), on_result, total=len(names))
    seconds = time.time() - start
    print(f"[rank {rank}] {done} images in {seconds:.1f}s")
    # the ONE collective of the sweep: per-rank (images, seconds) -> a summary on rank 0
    per_rank = D.gather_stats([done, seconds], device) if dist_on else [[float(done), seconds]]
    if rank == 0:
        total, slowest = sum(r[0] for r in per_rank), max(r[1] for r in per_rank)
        summary = {"world_size": world, "images": int(total), "seconds": slowest,
                   "images_per_s": total / slowest if slowest > 0 else 0.0,
                   "per_rank": [{"rank": i, "images": int(r[0]), "seconds": r[1]} for i, r in enumerate(per_rank)]}
        with open(os.path.join(args.output_dir, "summary.json"), "w") as f:
            json.dump(summary, f, indent=1)
        print(json.dumps(summary))
    return done


if __name__ == "__main__":
    a = get_args()
    r, w, _ = D.env_ranks()
    main(a, r, w)
    D.finish()
