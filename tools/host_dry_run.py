"""Multi-GPU host budget WITHOUT a multi-GPU node: N ranks over gloo on this box's host cores replay the REAL
per-image host work of the stage-1 driver (dvt_amd/stage1.py) with the GPU replaced by a sleep:

  * the index stream: np.random.randint(0, 769 * 1369, (1000, 2048)).astype(int32) (FitEngine.sample_indices; 2 M draws),
  * its copy into the 8-MB upload buffer (pinned on the GPU box; a plain preallocated buffer here),
  * the two output arrays (raw [37, 37, 768] + denoised [1, 37, 37, 768] fp32 = 8.4 MB) copied out of "pinned" host memory and
    written as .npy through misc.atomic_save_npy (temp file + rename) into a per-rank directory,
  * three more host threads per rank that spend their life blocked (extractor / fit / retire threads waiting on the HIP queue).

Each rank is paced at `--rate` images/s (3 = the measured per-GPU rate with margin) through dist.timed's barrier bracket, with
the per-rank CPU pinning of dist.pin_host_threads ON and OFF.  Reported per mode: CPU-seconds per image (process CPU time),
the wall time the host work of one image takes (what bounds the per-rank image rate), and whether the pace was held.

    python tools/host_dry_run.py --ranks 8 --images 30 --rate 3 --out profiles/r04/host_dry_run.json
"""
import argparse
import json
import os
import sys
import tempfile
import threading
import time

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, images, rate, pin, tmp, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    if not pin:
        os.environ["DVT_NO_AFFINITY"] = "1"
    from dvt_amd import dist as D
    from dvt_amd.utils import misc
    cpu = torch.device("cpu")
    D.init(cpu, world)  # pins (or not), joins the gloo rendezvous
    misc.fix_random_seeds(rank)
    n_rows, iters, B = 769 * 1369, 1000, 2048
    upload = np.empty((iters, B), np.int32)          # the (pinned) index upload buffer
    raw_pin = np.random.rand(37, 37, 768).astype(np.float32)
    den_pin = np.random.rand(1, 37, 37, 768).astype(np.float32)
    stop = threading.Event()
    blocked = [threading.Thread(target=stop.wait, daemon=True) for _ in range(3)]  # threads parked on a full HIP queue
    for t in blocked:
        t.start()
    host_wall = []

    def one_image(k):
        t0 = time.perf_counter()
        idx = np.random.randint(0, n_rows, (iters, B)).astype(np.int32)   # FitEngine.sample_indices
        np.copyto(upload, idx)                                            # -> upload buffer (H2D source)
        raw, den = raw_pin.copy(), den_pin.copy()                          # out of the D2H landing buffers
        misc.atomic_save_npy(os.path.join(tmp, f"r{rank}", "raw_features", f"{k % 4}.npy"), raw)
        misc.atomic_save_npy(os.path.join(tmp, f"r{rank}", "denoised_features", f"{k % 4}.npy"), den)
        host_wall.append(time.perf_counter() - t0)

    def run():
        t_next = time.perf_counter()
        for k in range(images):
            one_image(k)
            t_next += 1.0 / rate
            slack = t_next - time.perf_counter()
            if slack > 0:
                time.sleep(slack)  # the GPU is busy with the image; the host has nothing to do
        return images

    one_image(-1)  # warm (allocations, directory creation)
    host_wall.clear()
    c0 = time.process_time()
    n, elapsed, per_rank = D.timed(run, cpu)
    cpu_s = time.process_time() - c0
    stop.set()
    q.put({"rank": rank, "cpu_s_per_image": cpu_s / images, "host_wall_ms_mean": 1e3 * float(np.mean(host_wall)),
           "host_wall_ms_max": 1e3 * float(np.max(host_wall)), "elapsed_s": elapsed,
           "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None})
    D.finish()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--images", type=int, default=30)
    ap.add_argument("--rate", type=float, default=3.0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    report = {"host_logical_cores": os.cpu_count(), "ranks": a.ranks, "images_per_rank": a.images, "paced_rate_per_rank": a.rate,
              "work_per_image": "2 M np.random.randint draws + 8 MB index copy + 2 x 4.2 MB array copies + 2 atomic .npy writes "
                                f"({'tmpfs' if shm else 'tempfile dir'}), 3 blocked threads per rank", "modes": {}}
    for pin in (True, False):
        with tempfile.TemporaryDirectory(dir=shm) as tmp:
            ctx = mp.get_context("spawn")
            q = ctx.Queue()
            port = 29600 + os.getpid() % 1500 + (0 if pin else 7)
            procs = [ctx.Process(target=worker, args=(r, a.ranks, port, a.images, a.rate, pin, tmp, q)) for r in range(a.ranks)]
            for p in procs:
                p.start()
            res = sorted((q.get(timeout=600) for _ in procs), key=lambda d: d["rank"])
            for p in procs:
                p.join()
        ideal = a.images / a.rate
        report["modes"]["pinned" if pin else "unpinned"] = {
            "cpu_s_per_image_mean": float(np.mean([r["cpu_s_per_image"] for r in res])),
            "cpu_s_per_image_max": float(np.max([r["cpu_s_per_image"] for r in res])),
            "host_wall_ms_per_image_mean": float(np.mean([r["host_wall_ms_mean"] for r in res])),
            "host_wall_ms_per_image_worst": float(np.max([r["host_wall_ms_max"] for r in res])),
            "elapsed_s": res[0]["elapsed_s"], "ideal_s": ideal, "slowdown_vs_pace": res[0]["elapsed_s"] / ideal,
            "cpus_per_rank": res[0]["affinity"],
            "host_bound_images_per_s_per_rank": 1e3 / float(np.mean([r["host_wall_ms_mean"] for r in res])),
        }
        m = report["modes"]["pinned" if pin else "unpinned"]
        print(f"{'pinned  ' if pin else 'unpinned'}: {a.ranks} ranks x {a.rate} images/s on {os.cpu_count()} cores ({m['cpus_per_rank']} per rank): "
              f"{m['cpu_s_per_image_mean'] * 1e3:.0f} ms CPU per image (max {m['cpu_s_per_image_max'] * 1e3:.0f}), host work "
              f"{m['host_wall_ms_per_image_mean']:.0f} ms wall per image (worst {m['host_wall_ms_per_image_worst']:.0f}) = "
              f"{m['host_bound_images_per_s_per_rank']:.1f} images/s per rank host-bound; paced run {m['elapsed_s']:.2f} s vs "
              f"{ideal:.2f} s ideal ({(m['slowdown_vs_pace'] - 1) * 100:+.1f} %)", flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
