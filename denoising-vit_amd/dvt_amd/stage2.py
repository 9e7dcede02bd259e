"""Stage-2 driver: train the generalizable `Denoiser` on the (raw, denoised) feature pairs stage 1 wrote.

Mirrors the reference's main_denoiser.py (flags :25-79, loop :174-264): sqrt-scaled learning rate (:173),
`CosineScheduler` with 15 % linear warm-up from 0 (:179-186), MSE + (1 - cosine) loss (:213-217), AdamW
(:174-178), rank-0 checkpoints `{denoiser, optimizer, step}` + a `latest.pth` symlink (:239-264).

MI355X layout: one process per GPU (`python -m torch.distributed.run --nproc-per-node N -m dvt_amd.stage2 ...`).
Every rank runs forward + loss + backward natively (csrc/dvt_stage2.hip) on its own batch; the gradient arena is
ONE flat fp32 buffer (33 MB for ViT-B), summed across ranks by ONE RCCL all-reduce per step and scaled by
1/world inside the AdamW kernel -- the reference's DistributedDataParallel (:138-140) reduces the same gradients
in 25-MB buckets.  Feature pairs are read by a host thread pool into pinned buffers and uploaded on a side
stream while the previous step computes.  The reference's per-sample image decode (used only by its PCA
visualisation, :253-262) is not performed; `--vis_freq` is accepted and ignored.
"""
from __future__ import annotations

import argparse
import math
import os
import queue
import re
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
import torch.distributed as dist

from . import dist as D
from .models.online_denoiser import Denoiser
from .models.vit_wrapper import MODEL_LIST
from .utils import misc
from .vit import SPECS


# ---- host pieces restated from the reference ---------------------------------------------------------------
class CosineScheduler:
    """dvt/utils/misc.py:211-241: [freeze zeros | linear warm-up | half cosine], indexable by iteration."""

    def __init__(self, base_value, final_value, total_iters, warmup_iters=0, start_warmup_value=0, freeze_iters=0):
        self.final_value, self.total_iters = final_value, total_iters
        n_cos = total_iters - warmup_iters - freeze_iters
        k = np.arange(n_cos)
        self.schedule = np.concatenate((
            np.zeros((freeze_iters)),
            np.linspace(start_warmup_value, base_value, warmup_iters),
            final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * k / len(k)))))
        if len(self.schedule) != total_iters:
            raise ValueError("schedule pieces do not add up to total_iters")

    def __getitem__(self, it):
        return self.final_value if it >= self.total_iters else self.schedule[it]


def sampler_indices(n: int, world: int, rank: int, distributed: bool, epoch: int = 0):
    """Endless index stream of dvt/dataset/sampler.py: `InfiniteSampler` (0..n-1 repeated) when not
    distributed, else `DistributedInfiniteSampler`: the rank-strided subset i = rank, rank + world, ...,
    shuffled once by numpy's default_rng(epoch) and then cycled."""
    if not distributed:
        own = list(range(n))
    else:
        own = list(range(n))[rank::world]
        np.random.default_rng(epoch).shuffle(own)
    while True:
        yield from own


class PairedFeatureList:
    """dvt/dataset/paired_list_dataset.py:9-46 without the image: entry -> (original_feats, denoised_feats),
    `denoised = load(feat_root/<entry with .npy>)`, `original` = the same path with `denoised_features` ->
    `raw_features`, both `.squeeze()`d; an entry whose denoised file is missing is replaced by a random other
    entry (:31-32)."""

    def __init__(self, data_list: str, feat_root: str):
        with open(data_list) as f:
            self.entries = [line.strip().split(" ")[0] for line in f if line.strip()]
        self.feat_root = feat_root

    def __len__(self):
        return len(self.entries)

    def paths(self, index: int):
        entry = self.entries[index]
        ext = os.path.splitext(entry)[1]
        den = os.path.join(self.feat_root, entry.replace(f"{ext}", ".npy"))
        return den, den.replace("denoised_features", "raw_features")

    def __getitem__(self, index: int):
        for _ in range(1000):
            den, raw = self.paths(index)
            if os.path.exists(den):
                return np.load(raw).squeeze(), np.load(den).squeeze()
            index = int(np.random.randint(len(self.entries)))
        raise FileNotFoundError(f"no denoised feature files under {self.feat_root}")


class BatchFeeder:
    """Reads `batch_size` pairs per step with a thread pool into one of `depth` pinned buffer pairs and uploads
    them on a side stream; `next()` hands out device tensors [B, h, w, C] whose copy the compute stream waits for."""

    def __init__(self, dataset, indices, batch_size, shape, device, workers=8, depth=3):
        self.ds, self.it, self.bs, self.device = dataset, indices, batch_size, device
        self.pool = ThreadPoolExecutor(max(1, workers))
        self.copy_stream = torch.cuda.Stream(device) if device.type == "cuda" else None
        pin = device.type == "cuda"
        self.slots = [tuple(torch.empty((batch_size, *shape), dtype=torch.float32, pin_memory=pin) for _ in range(2))
                      for _ in range(depth)]
        self.dev = [tuple(torch.empty((batch_size, *shape), dtype=torch.float32, device=device) for _ in range(2))
                    for _ in range(depth)]
        self.events = [None] * depth
        self.free = queue.Queue()
        for i in range(depth):
            self.free.put(i)
        self.ready = queue.Queue(maxsize=depth)
        self.stop = False
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _load_one(self, slot, j, index):
        raw, den = self.ds[index]
        self.slots[slot][0][j].copy_(torch.from_numpy(np.ascontiguousarray(raw, dtype=np.float32)))
        self.slots[slot][1][j].copy_(torch.from_numpy(np.ascontiguousarray(den, dtype=np.float32)))

    def _run(self):
        try:
            while not self.stop:
                slot = self.free.get()
                if slot is None:
                    return
                idx = [next(self.it) for _ in range(self.bs)]
                list(self.pool.map(lambda a: self._load_one(slot, *a), enumerate(idx)))
                if self.copy_stream is not None:
                    with torch.cuda.stream(self.copy_stream):
                        for h, d in zip(self.slots[slot], self.dev[slot]):
                            d.copy_(h, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(self.copy_stream)
                    self.events[slot] = ev
                else:
                    for h, d in zip(self.slots[slot], self.dev[slot]):
                        d.copy_(h)
                self.ready.put(slot)
        except BaseException as e:  # surface reader failures in the training thread
            self.ready.put(e)

    def next(self):
        slot = self.ready.get()
        if isinstance(slot, BaseException):
            raise slot
        if self.events[slot] is not None:
            torch.cuda.current_stream(self.device).wait_event(self.events[slot])
        return slot, self.dev[slot]

    def release(self, slot, done_event=None):
        """The device buffers of `slot` may be overwritten once `done_event` (recorded after the step that
        consumed them) has completed."""
        if done_event is not None:
            done_event.synchronize()
        self.free.put(slot)

    def close(self):
        self.stop = True
        self.free.put(None)
        self.pool.shutdown(wait=False)


# ---- checkpoints -------------------------------------------------------------------------------------------
def optimizer_state(model: Denoiser, lr: float, weight_decay: float) -> dict:
    """`torch.optim.AdamW.state_dict()` layout over `model.parameters()` order, built from the flat moments."""
    eng = model.engine
    m, v = eng.views(eng.exp_avg), eng.views(eng.exp_avg_sq)
    names = [n for n, _ in model.named_parameters() if not n.startswith("vit.")]
    state = {i: {"step": torch.tensor(float(eng.step)), "exp_avg": m[n].detach().cpu().clone(),
                 "exp_avg_sq": v[n].detach().cpu().clone()} for i, n in enumerate(names)}
    group = {"lr": lr, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": weight_decay, "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(len(names)))}
    return {"state": state, "param_groups": [group]}


def save_checkpoint(log_dir: str, model: Denoiser, step: int, lr: float, weight_decay: float) -> str:
    """main_denoiser.py:239-264: ckpt_{step:06d}.pth + latest.pth symlink."""
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if "vit." not in k}
    path = f"{log_dir}/checkpoints/ckpt_{step:06d}.pth"
    torch.save({"denoiser": sd, "optimizer": optimizer_state(model, lr, weight_decay), "step": step}, path)
    latest = f"{log_dir}/checkpoints/latest.pth"
    try:
        os.remove(latest)
    except FileNotFoundError:
        pass
    os.symlink(os.path.abspath(path), latest)
    return path


def load_checkpoint(path: str, model: Denoiser) -> int:
    ck = torch.load(path, map_location="cpu", weights_only=False)
    model.load_state_dict(ck["denoiser"])
    eng = model.engine
    names = [n for n, _ in model.named_parameters() if not n.startswith("vit.")]
    m, v = eng.views(eng.exp_avg), eng.views(eng.exp_avg_sq)
    for i, n in enumerate(names):
        st = ck["optimizer"]["state"][i]
        m[n].copy_(st["exp_avg"].to(eng.device).reshape(m[n].shape))
        v[n].copy_(st["exp_avg_sq"].to(eng.device).reshape(v[n].shape))
        eng.step = int(st["step"])
    return int(ck["step"]) + 1


# ---- CLI ---------------------------------------------------------------------------------------------------
def get_args(argv=None):
    p = argparse.ArgumentParser("Train generalizable denoiser (MI355X)")
    p.add_argument("--model", type=str, default="vit_base_patch14_dinov2.lvd142m", choices=MODEL_LIST)
    p.add_argument("--num_blocks", type=int, default=1)
    p.add_argument("--data_root", type=str, default="data/imagenet")
    p.add_argument("--feat_root", type=str, default=None)
    p.add_argument("--data_list_path", type=str, default=None)
    p.add_argument("--input_size", type=int, default=518, nargs="+")
    p.add_argument("--auto_stride", action="store_true")
    p.add_argument("--stride_size", type=int, default=14)
    p.add_argument("--num_workers", default=8, type=int)
    p.add_argument("--batch_size", default=32, type=int, help="Batch size per GPU")
    p.add_argument("--num_vis_samples", default=8, type=int)
    p.add_argument("--num_iterations", default=40_000, type=int)
    p.add_argument("--weight_decay", type=float, default=1e-5)
    p.add_argument("--blr", type=float, default=2.0e-04)
    p.add_argument("--min_lr", type=float, default=1.0e-06)
    p.add_argument("--warmup_iters", type=int, default=50_000)
    p.add_argument("--output_root", default="./work_dirs/", type=str)
    p.add_argument("--save_freq", default=5000, type=int)
    p.add_argument("--vis_freq", default=5000, type=int)
    p.add_argument("--project", default="denosing-vit", type=str)
    p.add_argument("--run_name", default="debug", type=str)
    p.add_argument("--seed", default=42, type=int)
    p.add_argument("--device", default="cuda")
    p.add_argument("--resume", default=None, type=str, help="checkpoint to continue from (not in the reference)")
    p.add_argument("--log_freq", default=50, type=int)
    args = p.parse_args(argv)
    if isinstance(args.input_size, int):
        args.input_size = (args.input_size, args.input_size)
    elif len(args.input_size) == 1:
        args.input_size = (args.input_size[0], args.input_size[0])
    if args.auto_stride:
        args.stride_size = int(re.search(r"patch(14|16)", args.model).group(1))
    if args.stride_size in (8, 16) and args.input_size[0] == 518:
        args.input_size = (512, 512)
    if args.input_size[0] % args.stride_size or args.input_size[1] % args.stride_size:
        raise SystemExit("input size must be divisible by stride_size")
    return args


def model_geometry(args):
    """feat_dim and noise-map size as main_denoiser.py:112-118 derives them from the (unloaded) ViT."""
    if args.model not in SPECS:
        raise NotImplementedError(f"{args.model}: only the DINOv2 S/B/L (+reg4) feature maps are supported")
    spec = SPECS[args.model]
    return spec.dim, (args.input_size[0] - spec.patch) // args.stride_size + 1, \
        (args.input_size[1] - spec.patch) // args.stride_size + 1


def train(args, rank: int, world: int, device: torch.device, model_factory=None) -> dict:
    """The loop of main_denoiser.py:188-264.  `model_factory` lets the CPU tests inject a stand-in model with
    the engine interface (the product model needs a HIP device)."""
    distributed = world > 1
    log_dir = os.path.join(args.output_root, args.project, args.run_name)
    if rank == 0:
        os.makedirs(f"{log_dir}/checkpoints", exist_ok=True)
    misc.fix_random_seeds(args.seed)
    feat_dim, pos_h, pos_w = model_geometry(args)
    if model_factory is None:
        model = Denoiser(noise_map_height=pos_h, noise_map_width=pos_w, feat_dim=feat_dim, vit=None,
                         num_blocks=args.num_blocks, device=device)
    else:
        model = model_factory(pos_h, pos_w, feat_dim, args.num_blocks, device)
    eng = model.engine
    if distributed:  # DistributedDataParallel broadcasts rank 0's parameters at construction (:138-140)
        dist.broadcast(eng.params, src=0)
    start = load_checkpoint(args.resume, model) if args.resume else 0

    ds = PairedFeatureList(args.data_list_path, args.feat_root)
    feeder = BatchFeeder(ds, sampler_indices(len(ds), world, rank, distributed), args.batch_size,
                         (pos_h, pos_w, feat_dim), device, workers=args.num_workers)
    lr_base = args.blr * math.sqrt(args.batch_size * world / 256)
    sched = CosineScheduler(lr_base, args.min_lr, args.num_iterations,
                            warmup_iters=int(args.num_iterations * 0.15), start_warmup_value=0)
    history, t_log, last = [], time.time(), None
    pending = []  # (slot, event) of steps whose input buffers are still in flight
    # The reference tests the loss on EVERY step (main_denoiser.py:223-226) with a host sync.  Here a sticky device-side
    # flag collects "some step's loss was not finite" without a sync; it is reduced over ranks (MAX) and read wherever
    # the host looks anyway -- log steps and BEFORE every checkpoint -- so that every rank aborts together (no rank is
    # left blocked in all_reduce) and no NaN-poisoned parameters or AdamW moments are ever written to a checkpoint.
    bad = torch.zeros((), device=device, dtype=torch.float32)
    bad_step = torch.full((), float("inf"), device=device, dtype=torch.float32)  # +inf = not seen on this rank

    def raise_if_bad(step):
        flag = torch.stack([bad, -bad_step])  # one MAX reduction: any rank bad, and the EARLIEST step over the ranks
        if distributed:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        b, neg_at = flag.cpu().tolist()
        if b != 0.0:
            raise FloatingPointError(f"loss is not finite (first seen at step {int(-neg_at)}, detected at step "
                                     f"{step}), stopping training")

    try:
        for step in range(start, args.num_iterations):
            lr = float(sched[step])
            slot, (orig, den) = feeder.next()
            loss = model.training_step(orig, den)
            nf = (~torch.isfinite(loss.detach()[0])).float()
            bad_step = torch.where((bad == 0) & (nf != 0), torch.full_like(bad_step, float(step)), bad_step)
            bad = torch.maximum(bad, nf)
            if distributed:
                dist.all_reduce(eng.grads)  # SUM; the mean over ranks is taken inside the AdamW kernel
            eng.adamw_step(lr, args.weight_decay, grad_scale=1.0 / world)
            ev = None
            if device.type == "cuda":
                ev = torch.cuda.Event()
                ev.record()
            pending.append((slot, ev))
            if len(pending) > 1:
                feeder.release(*pending.pop(0))
            is_log = step % args.log_freq == 0 or step == args.num_iterations - 1
            is_save = step % args.save_freq == 0 or step == args.num_iterations - 1
            if is_log or is_save:
                raise_if_bad(step)  # collective: the same steps on every rank
            if is_log:
                vals = loss.detach().cpu().tolist()  # the only other host sync of the loop
                now = time.time()
                last = {"step": step, "loss": vals[0], "l2_loss": vals[1], "cosine_similarity_loss": vals[2], "lr": lr,
                        "iter_time": (now - t_log) / max(1, args.log_freq if step else 1)}
                t_log = now
                history.append(last)
                if rank == 0:
                    print("Train  [{step}/{n}]  loss: {loss:.6f}  l2_loss: {l2_loss:.6f}  cosine_similarity_loss: "
                          "{cosine_similarity_loss:.6f}  lr: {lr:.3e}  iter_time: {iter_time:.4f}".format(
                              n=args.num_iterations, **last), flush=True)
            if rank == 0 and is_save:
                save_checkpoint(log_dir, model, step, lr, args.weight_decay)
    finally:
        feeder.close()
    return {"log_dir": log_dir, "history": history, "model": model}


def main(argv=None):
    args = get_args(argv)
    rank, world, local = D.env_ranks()
    device = torch.device(args.device, local) if args.device == "cuda" else torch.device(args.device)
    if device.type == "cuda":
        torch.cuda.set_device(device)
    D.init(device, world)
    try:
        train(args, rank, world, device)
    finally:
        D.finish()


if __name__ == "__main__":
    sys.exit(main())
