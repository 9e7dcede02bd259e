#!/usr/bin/env python
"""bench.py -- stage-1 images denoised / second on N MI355X (BASELINE.json metric).

One "step" = one image of BASELINE.json configs[1]: DINOv2 ViT-B/14 features of 768 synthetic
518x518 views + the original (769 forwards, HIP bf16-MFMA extractor), a fresh neural field + 1000
fused Adam steps (B=2048, L=16, F=8), the final F(lattice) inference and the D2H copy of the two
output arrays.  Inputs (views, coordinates) are resident in HBM before the timed region.
N>1: one rank per GPU, images are independent units -> weak scaling, no collective in the loop
(barrier + max-over-ranks timing + ONE gather of per-rank counters, dvt_amd/dist.py).

`value` is the reference's `--dtype bfloat16` mode end to end (bf16 ViT, bf16-operand fit MLP);
`value_fp32_fit` repeats the timed region with fp32-operand fit GEMMs (the reference's default
precision for the fit; the extractor of this line stays bf16); `value_fp32` is the reference's DEFAULT precision end to
end (`--dtype float32`: exact-fp32 matrix cores in extractor and fit) and `value_fp32_matmul_high` the same with the opt-in
`--fp32_matmul high` (torch's float32 matmul precision "high": the extractor's matrix products as bf16x3, ~1e-5 relative,
fp32 accumulation), three pipelined images each inside the same timed bracket.  `parity` is measured in the same
process: the HIP chain against the CPU oracle chain on the sample the CPU baseline is timed on.

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

# HBM-side bytes per launch from PMC counters: separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over
# tools/pmc_target.py at THIS bench's batching (110-view extractor launches + 60 fit steps; `bash tools/gpu.sh pmc`),
# reduced by tools/pmc_traffic.py to profiles/<round>/pmc_traffic.json: (FETCH_SIZE x 2 [gfx950 wide-load correction,
# MI355X_MICROARCH.md HBM section] + WRITE_SIZE) x 1024 B / launches, calibrated there on layernorm (reads + writes =
# its algorithmic bytes).  The newest committed file wins; without one `traffic` is null.
def _pmc_traffic():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")))
    if not files:
        return {}, None
    try:
        with open(files[-1]) as fh:
            doc = json.load(fh)
        return ({k: v.get("bytes_per_launch") for k, v in doc.get("probes", {}).items()}, os.path.relpath(files[-1], ROOT),
                doc.get("extract_launch_views"))
    except Exception:
        return {}, None, None


# PMC_TRAFFIC_VIEWS: views of the ONE extractor launch the PMC passes profiled (the ViT kernels' bytes are proportional to the
# views of a launch: the bench line scales them to the launches it actually made, 769 views -> 398 + 371)
PMC_TRAFFIC_BYTES_PER_LAUNCH, PMC_TRAFFIC_SOURCE, PMC_TRAFFIC_VIEWS = _pmc_traffic()
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16
MFMA_F32_PEAK_TF = 157.3    # f32-input MFMA
MFMA_BF16_SUSTAINED_TF = 2069.0  # measured: pure-MFMA loop, random operands, at the socket power cap (profiles/r03)
# SURVEY.md 8(d) image-level ceilings for ViT-B/14: extractor 233.1 TFLOP / 2.5 PF/s = 93 ms, fit 0.549 TB / 8 TB/s =
# 69 ms per image -> serial 1 / (93 + 69 ms) = 6.2 images/s, phases overlapped across images 1 / max = 10.7 images/s
IMAGE_CEILINGS = {"vit_base_patch14_dinov2.lvd142m": (6.2, 10.7)}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20, help="images timed per rank (default = the flags the round-end driver passes)")
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--model", default="vit_base_patch14_dinov2.lvd142m")
    p.add_argument("--num-iters", type=int, default=1000)
    p.add_argument("--warmup-iters", type=int, default=100)
    p.add_argument("--views", type=int, default=768)
    p.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (baseline + parity)")
    p.add_argument("--no-probes", action="store_true")
    p.add_argument("--no-fp32-fit", action="store_true", help="skip the second timed region (fp32-operand fit)")
    p.add_argument("--no-vit-large", action="store_true",
                   help="skip the BASELINE configs[2] leg (ViT-L/14 + 4 concurrent fits, value_vit_large_k4)")
    p.add_argument("--no-stage2", action="store_true",
                   help="skip the BASELINE configs[4] leg (stage-2 Denoiser training step, value_stage2_samples_per_s)")
    p.add_argument("--vit-cus-per-32", type=int, default=32,
                   help="CUs (of every 32) the extractor stream may use; <32 keeps some free for the fit")
    p.add_argument("--tune", type=str, default="", help="developer knobs: key=value,... for dvt_tune_set")
    p.add_argument("--fit-batch", type=int, default=4,
                   help="images whose fits share launches (dvt_fit_run_batched); 1 = one fit at a time.  4 (default since round "
                        "6): +0.9 % images/s at these flags on one box (2.957 / 2.960 against 2.931 / 2.932, "
                        "profiles/r06/r06c_*), +1.8 % at --steps 40 (profiles/r05/r05j_*), for 3.2 GB of feature store per "
                        "extra image in flight")
    p.add_argument("--fit-dtype", default="bfloat16", choices=["bfloat16", "float32"],
                   help="operand precision of the fit's MLP GEMMs for `value` (the reference's --dtype; bfloat16 = "
                        "its autocast mode, which the ViT of this bench always runs in)")
    p.add_argument("--pixel-bsz", type=int, default=2048,
                   help="developer experiment only: anything but 2048 is not BASELINE's workload")
    p.add_argument("--no-npy", action="store_true",
                   help="do not write the two .npy files per image (tmpfs) inside the timed region")
    p.add_argument("--save-root", default=None, help="where the timed region writes its .npy pairs (default: a "
                                                     "directory under /dev/shm, removed afterwards)")
    p.add_argument("--extract-launch-views", type=int, default=0,
                   help="cap on the views per extractor launch (0 = 400: 769 views -> 398 + 371)")
    p.add_argument("--rendezvous-only", action="store_true",
                   help="test hook: join the N-rank process group over gloo, run the timed bracket on no work, print one line")
    p.add_argument("--pipeline-depth", type=int, default=2,
                   help="images in flight per GPU (1 = strictly serial reference flow)")
    return p.parse_args()


def stage1_args(a):
    return SimpleNamespace(
        model=a.model, input_size=(518, 518), stride_size=14, layer_depth_ratio=1.0,
        num_views=a.views, num_iters=a.num_iters, warmup_iters=a.warmup_iters, n_levels=16,
        freeze_shared_artifacts_after=0.5, lr=0.01, min_lr=0.001, weight_decay=1e-5,
        extract_bsz=32, extract_launch_views=a.extract_launch_views, pixel_bsz=a.pixel_bsz, seed=0, vit_checkpoint=None,
        dtype=a.fit_dtype)


def _best_threads(fn):
    """The GPU box exposes many host threads; small torch ops get SLOWER with all of them.  Time one
    call of `fn` at a few thread counts and keep the fastest (the count is reported as `cores`)."""
    total = os.cpu_count() or 8
    best, best_t = None, float("inf")
    for n in sorted({min(total, c) for c in (8, 16, 32, 64, total)}):
        torch.set_num_threads(n)
        fn()
        t0 = time.perf_counter()
        fn()
        t = time.perf_counter() - t0
        if t < best_t:
            best, best_t = n, t
    torch.set_num_threads(best)
    return best


def cpu_baseline_and_parity(a, vit, device):
    """The oracle (CPU restatement of the reference path: fp32 ViT + pure-PyTorch hash-grid field +
    torch.optim.Adam; the reference itself has no CPU path: tiny-cuda-nn is CUDA-only) timed on this
    box's host cores on a bounded sample of the SAME workload -- 16 synthetic views + the original
    through the 12-block ViT-B/14 and 100 Adam steps at the full fit configuration --
    and extrapolated linearly.  The very same oracle outputs then serve as the checker of the HIP
    chain on that sample (`parity`): identical weights, views, initial parameters and index stream."""
    from dvt_amd import views as Vw
    from dvt_amd.fit import FitEngine, FitSettings
    from dvt_amd.models import NeuralFeatureField, SingleImageDenoiser
    from oracle import fit as ofit
    from oracle import vit as ovit
    from oracle.models import NeuralFeatureFieldOracle, SingleImageDenoiserOracle

    sd = vit._state_dict
    C = vit.n_output_dims
    # 16 views + the original, 100 steps across the phase switch: ~20 s of host work on 16 threads (round 2: 8 views, 20 steps)
    V, T, WARM, B, H = 16, 100, 10, a.pixel_bsz, 37
    views, coords = Vw.synthetic_views(V, (518, 518), H, H, device, seed=4242)
    views_c, coords_c = views.cpu(), coords.cpu()
    with torch.no_grad():
        cores = _best_threads(lambda: ovit.forward_features(sd, views_c[:1], 14, 14, n_blocks=2))
        t0 = time.perf_counter()
        feats_o = torch.cat([ovit.forward_features(sd, views_c[i:i + 1], 14, 14) for i in range(V + 1)])
        t_view = (time.perf_counter() - t0) / (V + 1)
    n_rows = (V + 1) * H * H
    idx = np.random.RandomState(0).randint(0, n_rows, (T, B)).astype(np.int32)

    def fresh():
        torch.manual_seed(0)
        return SingleImageDenoiserOracle(H, H, C, 11), NeuralFeatureFieldOracle(feat_dim=C, n_levels=16)

    d_w, f_w = fresh()
    ofit.fit_image(d_w, f_w, feats_o, coords_c, idx[:2], num_iters=2, warmup_iters=1)  # warm (allocations)
    del d_w, f_w
    d_o, f_o = fresh()
    init_d = {k: v.clone() for k, v in d_o.state_dict().items()}
    init_f = {k: v.clone() for k, v in f_o.state_dict().items()}
    t0 = time.perf_counter()
    ofit.fit_image(d_o, f_o, feats_o, coords_c, idx, num_iters=T, warmup_iters=WARM)
    t_step = (time.perf_counter() - t0) / T
    want = ofit.final_denoised_feats(d_o, f_o, feats_o, coords_c)[0]
    sec_per_image = t_view * (a.views + 1) + t_step * a.num_iters
    base = {"value": 1.0 / sec_per_image, "unit": "images/s", "cores": cores, "host_cores": os.cpu_count(),
            "kind": "port",
            "sample": f"{V + 1} fp32 ViT-B/14 views ({t_view:.2f} s each) + {T} fit steps at B={B}, L=16, 2^20 "
                      f"({t_step:.3f} s each) on {cores} torch threads (the fastest of a 2-block probe over 8/16/32/64/all of "
                      f"the box's {os.cpu_count()} logical host cores; the hosts are shared), extrapolated linearly to {a.views + 1} views + "
                      f"{a.num_iters} steps ({sec_per_image:.0f} s/image)"}
    # ---- the HIP chain on the same sample
    cos = torch.nn.functional.cosine_similarity
    with torch.no_grad():
        feats_h = vit.features_nhwc(views, 11)
    raw_cos = cos(feats_h.reshape(-1, C).cpu().double(), feats_o.reshape(-1, C).double(), dim=-1)
    par = {"sample": f"{V + 1} synthetic views, {T} Adam steps, identical weights / init / index stream; "
                     "HIP chain (bf16 ViT -> fit) vs CPU oracle chain (fp32 ViT -> fp32 fit)",
           "metric": "per-patch cosine of the saved tensor denoised_feats [37,37,768] (north-star bar >= 0.99)",
           "raw_features_cos_mean": float(raw_cos.mean()), "raw_features_cos_min": float(raw_cos.min()),
           # (this in-line sample is small and chaotic run to run; the parity EVIDENCE is the pair of committed oracle fixtures)
           "full_schedule": "the metric's literal configuration is held against committed CPU-oracle fixtures by pytest -m gpu: "
                            "tests/test_gpu_parity_full.py::test_chain_metric_configuration_vs_oracle_fixture -- the WHOLE chain, "
                            "768 crops + the original -> 12-block ViT-B/14 -> 1 052 761-row store -> 1000 Adam steps -> saved tensor "
                            "(tests/golden/chain769_c768.npz; profiles/r06/final/gpu_suite_summary.txt: per-patch cosine 0.99898 mean / "
                            "0.9902 min bf16 mode, 0.99927 / 0.9934 --dtype float32, next to the oracle's own 1e-6-perturbation "
                            "sensitivity 0.99941 / 0.9927) -- and ::test_fit_metric_configuration_769_views_vs_oracle_fixture -- the "
                            "1000-step fit alone on 769 synthetic views (0.99996 / 0.9971 bf16 mode, 0.99999 / 0.9989 fp32)"}
    f_h, d_h = NeuralFeatureField(feat_dim=C, n_levels=16), SingleImageDenoiser(H, H, C, 11)
    f_h.load_state_dict(init_f)
    d_h.load_state_dict(init_d)
    f_h, d_h = f_h.to(device), d_h.to(device)
    for mode in ("bfloat16", "float32"):
        s = FitSettings(feat_dim=C, num_iters=T, warmup_iters=WARM, pixel_bsz=B, mlp_dtype=mode)
        eng = FitEngine(s, n_rows, device)
        eng.load_modules(d_h, f_h)
        eng.fit(feats_h.reshape(-1, C), coords.reshape(-1, 2), idx, log_every=0)
        got = eng.infer(coords[-1]).cpu()
        c = cos(got.reshape(-1, C).double(), want.reshape(-1, C).double(), dim=-1)
        par[f"denoised_feats_cos_mean_{mode}_fit"] = float(c.mean())
        par[f"denoised_feats_cos_min_{mode}_fit"] = float(c.min())
        del eng
    return base, par


VIT_LARGE = "vit_large_patch14_dinov2.lvd142m"


def vit_large_leg(a, device, rank, D, V, Stage1, PretrainedViTWrapper):
    """BASELINE configs[2] inside the driver's line (VERDICT r4 missing #3): DINOv2 ViT-L/14 (24 blocks, C = 1024, 779.5 TFLOP per
    image), four images' neural fields fitted concurrently (dvt_fit_run_batched: shared launches, LDS-resident row images),
    the same pipelined driver, the same timed bracket: 4 untimed + 8 timed images, then one strictly serial image for the two
    reference timers."""
    la = argparse.Namespace(**vars(a))
    la.model, la.fit_batch = VIT_LARGE, 4
    sa = stage1_args(la)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vit_l = PretrainedViTWrapper(VIT_LARGE, stride=14, allow_random_init=True)
    st = Stage1(sa, device, vit=vit_l, depth=2, fit_batch=4)
    for k, slot in enumerate(st.slots):
        views, coords = V.synthetic_views(a.views, sa.input_size, st.pos_h, st.pos_w, device, seed=100 * rank + 50 + k)
        slot.views.copy_(views)
        slot.coords.copy_(coords)
        del views, coords

    def jobs(n):
        for k in range(n):
            yield k, (lambda slot: None)

    st.run(jobs(4), total=4)
    n, el, _ = D.timed(lambda: st.run(jobs(8), total=8), device)
    st.process(lambda slot: None)
    t = st.timings[-1]
    out = {"images_per_s": n / el, "images_timed": n, "ms_per_image": 1e3 * el / n, "model": VIT_LARGE, "fit_batch": 4,
           "flow": "pipelined, depth 2 groups of 4 images; no .npy writes in this leg",
           "t_extract_s_serial": t["t_extract"], "t_fit_s_serial_one_fit": t["t_fit"],
           "extract_launch_views": st.vit_launch_views(a.views + 1),
           "workload": f"BASELINE configs[2]: DINOv2 ViT-L/14 518x518, {a.views} views + original, {a.num_iters}-step fits at "
                       "C = 1024 (MLP 128 -> 512 -> 1024, h 1024 -> 256 -> 256 -> 1024), 4 concurrent neural fields per GPU",
           "frac_of_extractor_roof": (779.5e12 / (MFMA_BF16_PEAK_TF * 1e12)) / (el / n)}
    del st, vit_l
    torch.cuda.empty_cache()
    return out


def stage2_leg(device, D):
    """BASELINE configs[4] on ONE GPU (stage 2 had no driver-side number): the generalizable Denoiser's
    training step -- one timm Block on [32, 1369, 768] feature maps, MSE + cosine loss, exact-fp32 HIP forward + backward +
    AdamW (csrc/dvt_stage2.hip) -- on synthetic (raw, denoised) pairs resident in HBM: 3 untimed + 20 timed steps.  The
    data-parallel form adds ONE flat all-reduce of the gradient arena per step (tools/bench_stage2.py under torchrun)."""
    from dvt_amd.models import Denoiser
    batch, grid, dim, steps = 32, 37, 768, 20
    m = Denoiser(grid, grid, dim, None, True, 1, device=device, seed=0)
    g = torch.Generator(device=device).manual_seed(0)
    x = torch.randn(batch, grid, grid, dim, device=device, generator=g)
    t = torch.randn(batch, grid, grid, dim, device=device, generator=g)

    def step():
        loss = m.training_step(x, t)
        m.engine.adamw_step(1e-4, 1e-5, grad_scale=1.0)
        return loss

    for _ in range(3):
        step()

    def region():
        for _ in range(steps):
            step()
        return steps * batch

    n, el, _ = D.timed(region, device)
    C, T, F, H = dim, grid * grid, 4 * dim, dim // 64
    lin = 2.0 * batch * T * (3 * C * C + C * C + 2 * C * F)  # forward linear layers
    att = 4.0 * batch * H * T * T * 64                       # q k^T and P v
    flops = 3.0 * lin + (att + 1.5 * att * 2)                # backward: 2 x linear, 4 attention products (tools/bench_stage2.py)
    out = {"samples_per_s": n / el, "ms_per_step": 1e3 * el / steps, "batch": batch, "steps": steps,
           "achieved_tflops": flops * steps / el / 1e12, "frac_of_fp32_mfma_peak": flops * steps / el / 1e12 / MFMA_F32_PEAK_TF,
           "workload": "BASELINE configs[4] on one GPU: Denoiser (1 Block, dim 768, 37 x 37 tokens) training step, batch 32, "
                       "exact fp32, synthetic pairs; no DDP in this leg"}
    del m, x, t
    torch.cuda.empty_cache()
    return out


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(a):
    """`python bench.py --gpus N` as a PLAIN command (the shape the round-end driver uses; the reference's own multi-GPU form is N
    independent processes, sample_scripts/stage1.sh:8-20): with N > 1 and no torch.distributed.run environment the process
    re-launches itself as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py ...`
    (one rank per GPU, RCCL) and returns that job's exit code; rank 0 of the job prints the ONE JSON line.  It refuses only when
    fewer than N devices are visible.  Returns None when there is nothing to do (N = 1, or already a rank of a job)."""
    if a.gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return None
    if not a.rendezvous_only:
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_dev < a.gpus:
            print(f"bench.py: {a.gpus} GPUs requested, {n_dev} visible", file=sys.stderr, flush=True)
            return 2
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def rendezvous_only(a, D):
    """--rendezvous-only: the process-group half of an N-rank run and nothing else (no device, no library): every rank joins over
    gloo, passes the bench's own timed bracket (barrier, MAX over ranks, the one gather) with a unit of no work, rank 0 prints
    one JSON line.  What tests/test_sharding_gloo.py drives on the CPU-only build container to cover the self-launch path."""
    import torch.distributed as tdist
    rank, world, _ = D.env_ranks()
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    cpu = torch.device("cpu")
    D.init(cpu, world)
    n, el, per_rank = D.timed(lambda: 1, cpu)
    if rank == 0:
        print(json.dumps({"rendezvous_only": True, "n_gpus": world, "dist_world_size": tdist.get_world_size() if world > 1 else 1,
                          "dist_backend": tdist.get_backend() if world > 1 else None,
                          "per_rank": [{"rank": i, "units": int(r[0])} for i, r in enumerate(per_rank)]}), flush=True)
    D.finish()


def main():
    a = parse()
    rc = self_launch(a)
    if rc is not None:
        sys.exit(rc)
    from dvt_amd import dist as D
    if a.rendezvous_only:
        return rendezvous_only(a, D)
    rank, world, local = D.env_ranks()
    # `--gpus N` IS the world size: inside a job (self-launched above, or started by torch.distributed.run) the ranks check it
    assert world == a.gpus, (f"--gpus {a.gpus} but WORLD_SIZE={world}: launch N > 1 as `python bench.py --gpus {a.gpus}` or `python -m "
                             f"torch.distributed.run --nnodes=1 --nproc-per-node {a.gpus} --master-addr 127.0.0.1 --master-port P "
                             f"bench.py --gpus {a.gpus}`")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    D.init(device, world)
    import torch.distributed as tdist
    dist_world = tdist.get_world_size() if tdist.is_initialized() else 1
    assert dist_world == a.gpus, f"process group has {dist_world} ranks, --gpus {a.gpus}"
    dist_backend = tdist.get_backend() if tdist.is_initialized() else None  # "nccl" = RCCL on ROCm

    from dvt_amd import _lib
    from dvt_amd import views as V
    from dvt_amd.models import PretrainedViTWrapper
    from dvt_amd.stage1 import Stage1
    from dvt_amd.utils import misc

    _lib.lib()
    for kv in filter(None, a.tune.split(",")):
        k, v = kv.split("=")
        _lib.tune(int(k), int(v))  # (recorded as the user's: the driver's per-mode switches leave these keys alone)
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

    # one image includes its two .npy writes (main_img_denoising.py:131-146).  They happen on the
    # retiring thread of the pipeline, as in the driver (dvt_amd/stage1.py), into a tmpfs directory: the metric must
    # not depend on the GPU box's disk.  Layout and atomic rename are the driver's.
    save_root, npy_where = None, None
    if not a.no_npy:
        import shutil
        import tempfile
        shm_ok = os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (512 << 20)  # (docker default: 64 MB)
        save_root = a.save_root or tempfile.mkdtemp(prefix=f"dvt_bench_r{rank}_", dir="/dev/shm" if shm_ok else None)
        npy_where = "caller's --save-root" if a.save_root else ("tmpfs (/dev/shm)" if shm_ok else "tempfile default directory")
    written = [0, None]  # bytes, first error

    def write_pair(tag, raw_h, den_h):
        if save_root is None or written[1] is not None:
            return
        try:  # (two slots per rank are enough for a rate; a full disk must not take the bench line down)
            misc.atomic_save_npy(os.path.join(save_root, "raw_features", a.model, f"{tag % 2}.npy"), raw_h)
            misc.atomic_save_npy(os.path.join(save_root, "denoised_features", a.model, f"{tag % 2}.npy"), den_h)
            written[0] += int(raw_h.nbytes + den_h.nbytes)
        except OSError as exc:
            written[1] = repr(exc)

    def set_fit_dtype(mode):
        for e in st.engines:
            e.s.mlp_dtype = mode
            e.cfg.mlp_bf16 = int(mode == "bfloat16")

    st.run(jobs(a.warmup), total=a.warmup)
    probes = [] if a.no_probes else ["adam", "vit_gemm", "vit_attn", "fit_gemm", "grid"]
    # un-pipelined passes over one image, outside the timed region.  First with every probe OFF: the reference's two
    # timers (main_img_denoising.py:341, :355) -- round 4 took them with all five probes on and the ~15 k event pairs of a
    # fit inflated t_fit from 93 to 115 ms.  Then once more with the probes on: per-kernel durations WITHOUT a second
    # stream competing for the GPU.
    _lib.prof_enable([])
    st.process(lambda slot: None)
    split = st.timings[-1]
    _lib.prof_enable(probes)
    st.process(lambda slot: None)
    prof_iso = {n: _lib.prof_collect(n) for n in probes}
    # inside the timed region only the three heavy kernels are probed (~2.5 k event pairs per
    # image); probing all ~15 k small fit launches costs ~100 us/step (measured) and would
    # distort the metric
    probes = [n for n in probes if n in ("adam", "vit_gemm", "vit_attn")]
    _lib.prof_enable(probes)
    n_done, elapsed, per_rank = D.timed(lambda: st.run(jobs(a.steps), on_result=write_pair, total=a.steps), device)
    npy_bytes = written[0]
    assert n_done == a.steps
    prof = {n: _lib.prof_collect(n) for n in probes}
    _lib.prof_enable([])
    other = "float32" if a.fit_dtype == "bfloat16" else "bfloat16"
    second = None
    if not a.no_fp32_fit:  # the other fit precision, same timed bracket, fewer images
        k2 = max(2, a.steps // 3)
        set_fit_dtype(other)
        st.run(jobs(1), total=1)
        n2, el2, _ = D.timed(lambda: st.run(jobs(k2), on_result=write_pair, total=k2), device)
        st.process(lambda slot: None)
        second = {"images_per_s": world * n2 / el2, "images_timed_per_rank": n2,
                  "t_fit_s_serial": st.timings[-1]["t_fit"]}
        set_fit_dtype(a.fit_dtype)
    full_fp32 = None
    if not a.no_fp32_fit and world == 1 and a.model in IMAGE_CEILINGS:
        # the reference's DEFAULT precision end to end (--dtype float32: fp32 extractor + fp32 fit), the same pipelined
        # driver inside the same timed bracket, 5 images (exact-fp32 matrix cores: 1/16 of the bf16 rate; 3 images until
        # round 4 -- with ~2 s per image the pipeline's fill + drain weighed 4 % of a 3-image bracket)
        set_fit_dtype("float32")
        st.extract_dtype = "float32"
        st.run(jobs(1), total=1)  # builds the fp32 weight copies / workspace, warms the pipeline
        n3, el3, _ = D.timed(lambda: st.run(jobs(5), on_result=write_pair, total=5), device)
        st.process(lambda slot: None)
        t = st.timings[-1]
        full_fp32 = {"images_per_s": n3 / el3, "images_timed": n3, "flow": f"pipelined (depth {a.pipeline_depth})",
                     "t_extract_s_serial": t["t_extract"], "t_fit_s_serial": t["t_fit"]}
        # ... and the same with the extractor's matrix products as bf16x3 (`--fp32_matmul high`: torch's float32 matmul
        # precision "high", an opt-in the reference never sets -- reported beside value_fp32, never in its place)
        st.extract_matmul = "high"
        st.run(jobs(1), total=1)
        n4, el4, _ = D.timed(lambda: st.run(jobs(3), on_result=write_pair, total=3), device)
        st.process(lambda slot: None)
        t = st.timings[-1]
        full_fp32["matmul_high"] = {"images_per_s": n4 / el4, "images_timed": n4, "t_extract_s_serial": t["t_extract"],
                                    "t_fit_s_serial": t["t_fit"],
                                    "what": "--dtype float32 --fp32_matmul high: every matrix product of the extractor (linear "
                                            "layers, q.k^T, p.v) on the bf16 pipe over split operands (bf16x3, ~1e-5 relative per "
                                            "product, fp32 accumulation); LayerNorm, softmax, GELU, residual stream and the whole "
                                            "fit fp32"}
        st.extract_matmul = "highest"
        st.extract_dtype = "bfloat16"
        set_fit_dtype(a.fit_dtype)
    if save_root is not None and a.save_root is None:
        shutil.rmtree(save_root, ignore_errors=True)
    pipe_depth, launch_views = a.pipeline_depth, st.vit_launch_views(a.views + 1)
    large = None
    if not a.no_vit_large and world == 1 and a.model in IMAGE_CEILINGS and a.model != VIT_LARGE:
        del st  # ViT-B slots and workspace (~22 GB) are not needed any more
        torch.cuda.empty_cache()
        try:
            large = vit_large_leg(a, device, rank, D, V, Stage1, PretrainedViTWrapper)
        except Exception as exc:  # an extra leg must never take the bench line down
            large = {"error": repr(exc)}

    s2 = None
    if not a.no_stage2 and world == 1 and a.model in IMAGE_CEILINGS:
        try:
            s2 = stage2_leg(device, D)
        except Exception as exc:  # an extra leg must never take the bench line down
            s2 = {"error": repr(exc)}
    if rank == 0:
        out = {
            "metric": "stage-1 images denoised/sec (DINOv2-B/14, 518px, 1k Adam steps)",
            "value": world * a.steps / elapsed, "unit": "images/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # the arithmetic of `value`: bf16 operands / fp32 accumulation in the ViT always; the fit's MLP GEMMs by --fit-dtype
            "dtype": "bf16" if a.fit_dtype == "bfloat16" else "bf16 extractor + f32 fit MLP operands", "data": "synthetic",
            # self-verifying N: ranks of the torch.distributed process group that ran this line (None/1 = standalone)
            "dist_world_size": dist_world, "dist_backend": dist_backend,
            "config": {
                "workload": (("BASELINE configs[1]: DINOv2 ViT-B/14" if "vit_base" in a.model else
                              f"BASELINE configs[2]: DINOv2 ViT-L/14, {a.fit_batch} concurrent neural fields per GPU"
                              if "vit_large" in a.model else a.model) +
                             f" 518x518, {a.views} views + original, {a.num_iters}-step per-image fit (B={a.pixel_bsz}, "
                             "L=16, F=8, 2^20 hash) on 1 MI355X per rank; one image = 769 ViT forwards + the fit + the "
                             "final inference + D2H + " + ("its two .npy files" if not a.no_npy else "NO file writes")),
                "model": a.model, "views": a.views + 1, "num_iters": a.num_iters,
                "warmup_iters": a.warmup_iters, "pixel_bsz": a.pixel_bsz,
                "npy_writes": None if a.no_npy else {"where": f"{npy_where}, retiring thread, atomic rename",
                                                      "bytes_per_image": npy_bytes // max(1, a.steps),
                                                      "error": written[1]},
                "precision_mode": "reference --dtype bfloat16 (autocast) end to end: ViT bf16 MFMA / fp32 accumulate; "
                                  f"fit MLP GEMMs {a.fit_dtype} operands / fp32 accumulate + outputs; hash grid, "
                                  "losses, Adam: fp32.  The reference's DEFAULT is --dtype float32 (see value_fp32_fit)",
                "fit_dtype": a.fit_dtype,
                "fit_step": ("bfloat16: fused row kernel, hash-grid gradient gathered from per-step sorted lists, dense Adam "
                             "for coarse grid levels + MLPs + G, lazy Adam (same recurrence applied on demand, refresh every 32 "
                             "steps; replay arithmetic v_rcp / v_sqrt, 1 ulp each, unless --tune 10=1 selects the IEEE replay) for the fine "
                             "grid levels; float32: the same step with exact-fp32 operands (v_mfma_f32_16x16x4_f32 row kernel beside "
                             "the bf16 extractor, layer-by-layer exact-fp32 GEMMs beside the fp32 extractor; sorted lists; lazy Adam "
                             "with the IEEE replay)"),
                "extractor": "LayerNorm folded into the qkv / fc1 GEMMs (bf16 path); LayerNorm kernels in the fp32 path",
                "weights": "random init (no network for checkpoints)",
                "t_extract_s_serial": split["t_extract"], "t_fit_s_serial": split["t_fit"],
                "serial_timers": "one strictly serial image with every profiling probe OFF",
                "pipeline_depth": pipe_depth, "fit_batch": a.fit_batch,
                "extract_launch_views": launch_views,
                "images_per_rank": a.steps, "parallelism": f"images sharded over {world} GPU(s), no collective",
                "per_rank": [{"rank": i, "images": int(r[0]), "seconds": r[1]} for i, r in enumerate(per_rank)],
            },
        }
        if second is not None:
            key = "value_fp32_fit" if other == "float32" else "value_bf16_fit"
            out[key] = second["images_per_s"]
            out["config"][key + "_detail"] = second
        if s2 is not None:
            out["value_stage2_samples_per_s"] = s2.get("samples_per_s")
            out["config"]["value_stage2_detail"] = s2
        if large is not None:
            out["value_vit_large_k4"] = large.get("images_per_s")
            out["config"]["value_vit_large_k4_detail"] = large
        if full_fp32 is not None:
            out["value_fp32"] = full_fp32["images_per_s"]
            out["value_fp32_matmul_high"] = full_fp32["matmul_high"]["images_per_s"]
            out["config"]["value_fp32_detail"] = full_fp32

        def traffic_of(n):
            """PMC bytes per launch of probe `n`; the ViT kernels' bytes scaled from the profiled launch's views to the mean
            views of this run's launches (their traffic is proportional to the views of a launch)."""
            t = PMC_TRAFFIC_BYTES_PER_LAUNCH.get(n)
            if t is None or n not in ("vit_gemm", "vit_attn") or not PMC_TRAFFIC_VIEWS:
                return t
            lv = launch_views if isinstance(launch_views, (list, tuple)) else [launch_views]
            return t * (sum(lv) / len(lv)) / PMC_TRAFFIC_VIEWS

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
                               traffic=traffic_of(n),
                               launches=p["launches"], avg_us=1e3 * p["total_ms"] / p["launches"],
                               ms_per_image=p["total_ms"] / images,
                               # work = ALGORITHMIC flops / bytes (1370 tokens, K = 588 patch): frac follows from it
                               work_per_launch=p["work"] / p["launches"])
            return kern

        kern = kernel_table(prof, a.steps)
        if kern:
            # per-kernel tables: durations are HIP-EVENT brackets recorded by the library on the launch stream
            # (dvt_prof_*).  Inside the pipelined region a bracket also contains the time the kernel's workgroups wait
            # behind the other stream's, so `kernels` reads ~5-10 % above the rocprofv3 kernel-trace durations of the same
            # region (profiles/<round>/final/pipe_kernel_stats.txt); `kernels_isolated` (one serial image, one stream)
            # agrees with rocprofv3's serial table.
            out["kernels"] = kern  # inside the timed (pipelined) region
            out["kernels_isolated"] = kernel_table(prof_iso, 1)  # serial pass, one stream
            out["kernels_timing"] = "hipEvent brackets on the launch stream (see profiles/ for the rocprofv3 tables)"
            dom = max(kern, key=lambda k: kern[k]["ms_per_image"])
            out["roofline"] = {"kernel": dom, **{k: kern[dom][k] for k in
                                                 ("bound", "achieved", "peak", "unit", "frac", "traffic")},
                               "traffic_source": PMC_TRAFFIC_SOURCE}
            if kern[dom]["bound"] == "mfma" and kern[dom]["peak"] == MFMA_BF16_PEAK_TF:
                # `peak` is the 2.5 PF/s the contract names (2.4 GHz).  Under the 1.4 kW socket cap this chip sustains
                # 2.03 GHz on random operands: a PURE bf16 MFMA loop measures 2069 TF/s (tools/probes/mfma_power.hip,
                # profiles/r03/r03c_power_probe_mfma_only.txt) -- the fraction against THAT is the structural one.
                out["roofline"]["peak_sustained_measured"] = MFMA_BF16_SUSTAINED_TF
                out["roofline"]["frac_of_sustained"] = kern[dom]["achieved"] / MFMA_BF16_SUSTAINED_TF
        if a.model in IMAGE_CEILINGS and a.num_iters == 1000 and a.views == 768:
            ser, ovl = IMAGE_CEILINGS[a.model]
            per_gpu = out["value"] / world
            out["image_level"] = {"per_gpu_images_per_s": per_gpu, "ceiling_serial": ser, "ceiling_overlapped": ovl,
                                  "frac_of_serial_ceiling": per_gpu / ser, "frac_of_overlapped_ceiling": per_gpu / ovl,
                                  "definition": "SURVEY.md 8(d): 233.1 TFLOP / 2.5 PF/s + 0.549 TB / 8 TB/s per image"}
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity(a, vit, device)
            except Exception as exc:  # the baseline must never take the bench line down
                out["cpu_baseline"] = {"error": repr(exc)}
        # key order of the ONE line: the contract's fields, then every other scalar result, roofline and cpu_baseline, and only then
        # the long descriptive objects (a reader that shows the head of the line sees the numbers)
        head = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "dist_world_size", "dist_backend", "value_fp32", "value_fp32_fit", "value_fp32_matmul_high",
                "value_vit_large_k4", "value_stage2_samples_per_s", "roofline", "cpu_baseline"]
        out = {**{k: out[k] for k in head if k in out}, **{k: v for k, v in out.items() if k not in head}}
        print(json.dumps(out), flush=True)
    D.finish()


if __name__ == "__main__":
    main()
