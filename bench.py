#!/usr/bin/env python
"""bench.py -- stage-1 images denoised / second on N MI355X (BASELINE.json metric).

One "step" = one image of BASELINE.json configs[1]: DINOv2 ViT-B/14 features of 768
synthetic 518x518 views + the original (769 forwards, HIP bf16-MFMA extractor), a fresh
neural field + 1000 fused Adam steps (B=2048, L=16, F=8, fp32), the final F(lattice)
inference and the D2H copy of the two output arrays.  Inputs (views, coordinates) are
resident in HBM before the timed region.  N>1: one rank per GPU, images are independent
units -> weak scaling, no collective in the loop (barrier + max-over-ranks timing only).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time
import warnings
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "denoising-vit_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# HBM bytes per launch from PMC counters, collected in separate rocprofv3 --pmc passes
# (profiles/r01d_pmc/{fetch,write}.txt; tools/gpu_pmc.sh on a 128-view batch + 60 fit steps):
# (FETCH_SIZE x 2 [gfx950 wide-load correction, MI355X_MICROARCH.md HBM section] + WRITE_SIZE)
# x 1024 B / launches.  Calibration: layernorm reads 830.6 MB/launch = its algorithmic 830 MB.
# vit_gemm = launch-weighted mean of the qkv (1961 MB), proj / fc2 (2080 MB) and fc1 (2364 MB) GEMMs.
PMC_TRAFFIC_BYTES_PER_LAUNCH = {"vit_gemm": 2121.0e6, "vit_attn": 1108.3e6, "adam": 517.3e6,
                                "fit_gemm": 48.4e6, "grid": None}
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16
MFMA_F32_PEAK_TF = 157.3    # f32-input MFMA


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=6, help="images timed per rank")
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--model", default="vit_base_patch14_dinov2.lvd142m")
    p.add_argument("--num-iters", type=int, default=1000)
    p.add_argument("--warmup-iters", type=int, default=100)
    p.add_argument("--views", type=int, default=768)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-probes", action="store_true")
    p.add_argument("--vit-cus-per-32", type=int, default=32,
                   help="CUs (of every 32) the extractor stream may use; <32 keeps some free for the fit")
    p.add_argument("--tune", type=str, default="", help="developer knobs: key=value,... for dvt_tune_set")
    p.add_argument("--fit-batch", type=int, default=1,
                   help="images whose fits share launches (dvt_fit_run_batched); 1 = one fit at a time "
                        "(measured: 1.94 / 1.93 / 1.87 images/s at 1 / 2 / 4 -- the step is throughput-, "
                        "not launch-latency-bound)")
    p.add_argument("--fit-dtype", default="bfloat16", choices=["bfloat16", "float32"],
                   help="operand precision of the fit's MLP GEMMs (the reference's --dtype; bfloat16 = its "
                        "autocast mode, which the ViT of this bench always runs in)")
    p.add_argument("--pixel-bsz", type=int, default=2048,
                   help="developer experiment only: anything but 2048 is not BASELINE's workload")
    p.add_argument("--pipeline-depth", type=int, default=2,
                   help="images in flight per GPU (1 = strictly serial reference flow)")
    return p.parse_args()


def stage1_args(a):
    return SimpleNamespace(
        model=a.model, input_size=(518, 518), stride_size=14, layer_depth_ratio=1.0,
        num_views=a.views, num_iters=a.num_iters, warmup_iters=a.warmup_iters, n_levels=16,
        freeze_shared_artifacts_after=0.5, lr=0.01, min_lr=0.001, weight_decay=1e-5,
        extract_bsz=128, pixel_bsz=a.pixel_bsz, seed=0, vit_checkpoint=None, dtype=a.fit_dtype)


def cpu_baseline(a):
    """The oracle (CPU restatement of the reference path: fp32 ViT + pure-PyTorch hash-grid
    field + torch.optim.Adam) timed on this box's host cores on a bounded sample and
    extrapolated linearly (the reference itself has no CPU path: tiny-cuda-nn is CUDA-only)."""
    from dvt_amd.vit import SPECS, random_state_dict
    from oracle import fit as ofit
    from oracle import vit as ovit
    from oracle.models import NeuralFeatureFieldOracle, SingleImageDenoiserOracle

    cores = torch.get_num_threads()
    spec = SPECS[a.model]
    sd = random_state_dict(spec.dim, spec.depth, 14, 1370, seed=0)
    img = torch.randn(1, 3, 518, 518)
    with torch.no_grad():
        ovit.forward_features(sd, img, 14, 14, n_blocks=1)  # warm
        t0 = time.perf_counter()
        n_views = 3
        for _ in range(n_views):
            ovit.forward_features(sd, img, 14, 14)
        t_view = (time.perf_counter() - t0) / n_views
    C, H, W, V = spec.dim, 37, 37, 8
    torch.manual_seed(0)
    feats, xy = torch.randn(V, H, W, C), torch.rand(V, H, W, 2)
    f_o = NeuralFeatureFieldOracle(feat_dim=C, n_levels=16)
    d_o = SingleImageDenoiserOracle(H, W, C, spec.depth - 1)
    T = 6  # 3 steps of each phase (switch at int(0.5*T) = 3)
    idx = np.random.RandomState(0).randint(0, V * H * W, (T, 2048))
    t0 = time.perf_counter()
    ofit.fit_image(d_o, f_o, feats, xy, idx, num_iters=T, warmup_iters=1)
    t_step = (time.perf_counter() - t0) / T
    sec_per_image = t_view * (a.views + 1) + t_step * a.num_iters
    return {"value": 1.0 / sec_per_image, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n_views} fp32 ViT views ({t_view:.2f} s each) + {T} fit steps "
                      f"({t_step:.2f} s each), extrapolated linearly to {a.views + 1} views + "
                      f"{a.num_iters} steps ({sec_per_image:.0f} s/image)"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    from dvt_amd import _lib
    from dvt_amd import views as V
    from dvt_amd.models import PretrainedViTWrapper
    from dvt_amd.stage1 import Stage1
    from dvt_amd.utils import misc

    _lib.lib()
    for kv in filter(None, a.tune.split(",")):
        k, v = kv.split("=")
        _lib.check(_lib.lib().dvt_tune_set(int(k), int(v)), f"dvt_tune_set({k},{v})")
    misc.fix_random_seeds(rank)
    sa = stage1_args(a)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # random-init weights: there is no network for checkpoints
        vit = PretrainedViTWrapper(a.model, stride=14, allow_random_init=True)
    st = Stage1(sa, device, vit=vit, depth=a.pipeline_depth, vit_cus_per_32=a.vit_cus_per_32,
                fit_batch=a.fit_batch)
    for k, slot in enumerate(st.slots):  # inputs resident in HBM before the timed region
        views, coords = V.synthetic_views(a.views, sa.input_size, st.pos_h, st.pos_w, device,
                                          seed=100 * rank + k)
        slot.views.copy_(views)
        slot.coords.copy_(coords)
        del views, coords

    def jobs(n):
        for k in range(n):
            yield k, (lambda slot: None)  # views already resident

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    st.run(jobs(a.warmup))
    probes = [] if a.no_probes else ["adam", "vit_gemm", "vit_attn", "fit_gemm", "grid"]
    # un-pipelined pass over one image, outside the timed region: the reference's two timers
    # and per-kernel durations WITHOUT a second stream competing for the GPU
    _lib.prof_enable(probes)
    st.process(lambda slot: None)
    split = st.timings[-1]
    prof_iso = {n: _lib.prof_collect(n) for n in probes}
    # inside the timed region only the three heavy kernels are probed (~2.5 k event pairs per
    # image); probing all ~15 k small fit launches costs ~100 us/step (measured) and would
    # distort the metric
    probes = [n for n in probes if n in ("adam", "vit_gemm", "vit_attn")]
    _lib.prof_enable(probes)
    barrier()
    t0 = time.perf_counter()
    n_done = st.run(jobs(a.steps))
    barrier()
    elapsed = time.perf_counter() - t0
    assert n_done == a.steps
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = {n: _lib.prof_collect(n) for n in probes}
    _lib.prof_enable([])
    t_ext, t_fit = split["t_extract"] * a.steps, split["t_fit"] * a.steps

    if rank == 0:
        out = {
            "metric": "stage-1 images denoised/sec (DINOv2-B/14, 518px, 1k Adam steps)",
            "value": world * a.steps / elapsed, "unit": "images/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: DINOv2 ViT-B/14 518x518, 768 views + original, "
                            "1k-step per-image fit (B=2048, L=16, F=8, 2^20 hash) on 1 MI355X per rank",
                "model": a.model, "views": a.views + 1, "num_iters": a.num_iters,
                "warmup_iters": a.warmup_iters, "pixel_bsz": a.pixel_bsz,
                "arithmetic": ("ViT: bf16 MFMA / fp32 accumulate; fit MLP GEMMs: " +
                               ("bf16 operands / fp32 accumulate + outputs (reference --dtype bfloat16 autocast)"
                                if a.fit_dtype == "bfloat16" else "fp32-operand MFMA") +
                               "; hash grid, losses, Adam: fp32"),
                "fit_dtype": a.fit_dtype,
                "weights": "random init (no network for checkpoints)",
                "t_extract_s_serial": t_ext / a.steps, "t_fit_s_serial": t_fit / a.steps,
                "pipeline_depth": a.pipeline_depth, "fit_batch": a.fit_batch,
                "images_per_rank": a.steps, "parallelism": f"images sharded over {world} GPU(s), no collective",
            },
        }
        def kernel_table(pr, images):
            kern = {}
            for n, p in pr.items():
                if p["launches"] == 0:
                    continue
                sec = p["total_ms"] * 1e-3
                if n in ("adam", "grid"):
                    kern[n] = {"bound": "hbm", "achieved": p["work"] / sec / 1e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s"}
                else:
                    peak = (MFMA_F32_PEAK_TF if n == "fit_gemm" and a.fit_dtype == "float32"
                            else MFMA_BF16_PEAK_TF)
                    kern[n] = {"bound": "mfma", "achieved": p["work"] / sec / 1e12, "peak": peak,
                               "unit": "TFLOP/s"}
                kern[n].update(frac=kern[n]["achieved"] / kern[n]["peak"],
                               traffic=PMC_TRAFFIC_BYTES_PER_LAUNCH.get(n),
                               launches=p["launches"], avg_us=1e3 * p["total_ms"] / p["launches"],
                               ms_per_image=p["total_ms"] / images)
            return kern

        kern = kernel_table(prof, a.steps)
        if kern:
            dom = max(kern, key=lambda k: kern[k]["ms_per_image"])
            out["roofline"] = {"kernel": dom, **{k: kern[dom][k] for k in
                                                 ("bound", "achieved", "peak", "unit", "frac", "traffic")}}
            out["kernels"] = kern  # inside the timed (pipelined) region
            out["kernels_isolated"] = kernel_table(prof_iso, 1)  # serial pass, one stream
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(a)
            except Exception as exc:  # the baseline must never take the bench line down
                out["cpu_baseline"] = {"error": repr(exc)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
