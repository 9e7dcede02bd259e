"""Generate tests/golden/stage2_host.npz by IMPORTING THE REFERENCE's stage-2 host logic
(/root/reference, build container only):

    python tests/golden/make_stage2_golden.py

Pins `CosineScheduler` (dvt/utils/misc.py:211-241) at the configuration main_denoiser.py:179-186 builds and at a
short one, and the index streams of `InfiniteSampler` / `DistributedInfiniteSampler` (dvt/dataset/sampler.py).
The model (timm Block) is an absent third party and cannot be pinned this way.
"""
import importlib.util
import itertools
import math
import os

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    misc = load("dvt/utils/misc.py", "dvt_ref_misc")
    sampler = load("dvt/dataset/sampler.py", "dvt_ref_sampler")
    out = {}
    # main_denoiser.py:173-186 with the defaults (blr 2e-4, batch 32, 8 GPUs, 40k iterations)
    lr = 2.0e-4 * math.sqrt(32 * 8 / 256)
    s = misc.CosineScheduler(base_value=lr, final_value=1.0e-6, total_iters=40_000,
                             warmup_iters=int(40_000 * 0.15), start_warmup_value=0)
    out["sched_default_every100"] = np.asarray([s[i] for i in range(0, 40_100, 100)], dtype=np.float64)
    s = misc.CosineScheduler(base_value=3e-4, final_value=1e-6, total_iters=40, warmup_iters=6, start_warmup_value=0)
    out["sched_short"] = np.asarray([s[i] for i in range(45)], dtype=np.float64)
    data = list(range(11))
    out["infinite_11"] = np.asarray(list(itertools.islice(iter(sampler.InfiniteSampler(data)), 30)))
    for world in (2, 3):
        for rank in range(world):
            it = iter(sampler.DistributedInfiniteSampler(data, num_replicas=world, rank=rank))
            out[f"dist_11_w{world}_r{rank}"] = np.asarray(list(itertools.islice(it, 20)))
    np.savez(os.path.join(OUT, "stage2_host.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
