"""CPU, world_size 2 over gloo: the REAL N > 1 code of the stage-1 driver and of bench.py.

`dvt_amd.stage1.main(args, rank, world, ...)` is driven with a host-only stand-in for the per-GPU
engine (so work-list slicing, sharding, resume-by-existence, the output layout, atomic writes and
the single end-of-run gather all execute), and `dvt_amd.dist.timed` -- the barrier / max-over-ranks
/ gather bracket bench.py uses -- runs over gloo exactly as it runs over RCCL on the GPUs.
(sample_scripts/stage1.sh:8-20 semantics: disjoint contiguous slices, no data-path collective.)"""
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODEL = "vit_base_patch14_dinov2.lvd142m"


def _fake_stage(rank):
    class FakeStage:  # what main() touches of Stage1: .vit.transformation, .pos_h/.pos_w, .run()
        def __init__(self, args, device):
            norm = SimpleNamespace(mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0))
            self.vit = SimpleNamespace(transformation=SimpleNamespace(transforms=[norm]))
            self.pos_h = self.pos_w = 2

        def run(self, jobs, on_result, total=None):
            n = 0
            for tag, _set_views in jobs:
                on_result(tag, np.full((2, 2, 4), rank, np.float32), np.full((1, 2, 2, 4), rank, np.float32))
                n += 1
            return n
    return FakeStage


def _worker(rank, world, port, tmp, n_images):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "denoising-vit_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from dvt_amd import dist as D
    from dvt_amd import stage1
    assert D.env_ranks() == (rank, world, rank)
    data_root = os.path.join(tmp, "data")
    lst = os.path.join(tmp, "list.txt")
    argv = ["--img_path", lst, "--data_root", data_root, "--save_root", os.path.join(tmp, "out"),
            "--output_dir", os.path.join(tmp, "work"), "--num_imgs", str(n_images + 5), "--model", MODEL]
    args = stage1.get_args(argv)
    cpu = torch.device("cpu")
    done = stage1.main(args, rank, world, stage_factory=_fake_stage(rank), device=cpu)
    # bench.py's bracket: barrier, K units, barrier, MAX over ranks, one gather
    n, elapsed, per_rank = D.timed(lambda: (time.sleep(0.05 * (rank + 1)), 3 + rank)[1], cpu)
    assert n == 3 + rank and len(per_rank) == world and [int(r[0]) for r in per_rank] == [3, 4]
    assert elapsed >= 0.1 and all(r[1] <= elapsed + 1e-6 for r in per_rank)  # the slowest rank sets the time
    np.save(os.path.join(tmp, f"done{rank}.npy"), np.array([done]))
    D.finish()


@pytest.mark.parametrize("n_images", [7])
def test_two_rank_sharded_sweep(tmp_path, n_images):
    names = [f"d{i % 3}/img{i}.jpg" for i in range(n_images)]
    (tmp_path / "list.txt").write_text("".join(f"{n} some-label\n" for n in names))
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path), n_images), nprocs=2, join=True)
    done = [int(np.load(tmp_path / f"done{r}.npy")[0]) for r in range(2)]
    assert done == [4, 3]  # rank 0 owns the first ceil(7/2) images
    den_dir = tmp_path / "out" / "denoised_features" / MODEL
    files = sorted(p.name for p in den_dir.rglob("*.npy"))
    assert files == sorted(f"img{i}.npy" for i in range(n_images))
    owner = [int(np.load(p)[0, 0, 0, 0]) for p in sorted(den_dir.rglob("*.npy"), key=lambda q: int(q.stem[3:]))]
    assert owner == [0, 0, 0, 0, 1, 1, 1]
    raw = np.load(tmp_path / "out" / "raw_features" / MODEL / "d0" / "img0.npy")
    assert raw.shape == (2, 2, 4)
    summary = json.loads((tmp_path / "work" / "summary.json").read_text())  # the end-of-run gather, rank 0
    assert summary["world_size"] == 2 and summary["images"] == n_images
    assert [r["images"] for r in summary["per_rank"]] == [4, 3]
    for r in range(2):
        lines = (tmp_path / "work" / f"timings_rank{r}.jsonl").read_text().splitlines()
        assert len(lines) == done[r]
    # rerun: everything is skipped (idempotent resume by file existence)
    mp.spawn(_worker, args=(2, port + 1, str(tmp_path), n_images), nprocs=2, join=True)
    assert [int(np.load(tmp_path / f"done{r}.npy")[0]) for r in range(2)] == [0, 0]
    assert json.loads((tmp_path / "work" / "summary.json").read_text())["images"] == 0


def test_ranks_pin_disjoint_host_cpu_slices():
    """dist.pin_host_threads (called by dist.init when world > 1): the ranks of one node take disjoint, contiguous
    slices of the CPUs and cap their torch pools -- 8 ranks x (4 blocked launch threads + numpy draws + torch pool)
    must not oversubscribe the same cores.  Run in child processes: the affinity of the test runner stays untouched."""
    import json
    import subprocess
    import sys
    n_cpu = len(os.sched_getaffinity(0))
    if n_cpu < 2:
        pytest.skip("needs >= 2 CPUs")
    # pinned TWICE (stage1.main then stage2.train call dist.init in one process): the second call must not slice the slice
    code = ("import os, sys, json; sys.path[:0] = [%r, %r]; from dvt_amd import dist as D; D.pin_host_threads(); "
            "info = dict(D.pin_host_threads()); "
            "import torch; info['affinity'] = sorted(os.sched_getaffinity(0)); info['threads'] = torch.get_num_threads(); "
            "print(json.dumps(info))") % (ROOT, os.path.join(ROOT, "denoising-vit_amd"))
    seen = []
    for r in range(2):
        env = dict(os.environ, LOCAL_RANK=str(r), LOCAL_WORLD_SIZE="2", WORLD_SIZE="2")
        env.pop("DVT_NO_AFFINITY", None)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout
        info = json.loads(out.strip().splitlines()[-1])
        assert info["pinned"] and len(info["affinity"]) == n_cpu // 2 and info["threads"] <= 8
        seen.append(set(info["affinity"]))
    assert not (seen[0] & seen[1])
    # a launcher that exports NO local world size at all: nothing is pinned (WORLD_SIZE may span several hosts; on this
    # CPU-only host it also exceeds the device count) -- and it says so once
    env = dict(os.environ, LOCAL_RANK="1", WORLD_SIZE="16")
    for k in ("LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "SLURM_NTASKS_PER_NODE", "MPI_LOCALNRANKS"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout
    info = json.loads(out.strip().splitlines()[-1])
    assert not info["pinned"] and len(info["affinity"]) == n_cpu
    assert out.count("host threads are NOT pinned") == 1
    # mpirun / srun: their own names for the ranks of this host are honoured
    for extra in ({"OMPI_COMM_WORLD_LOCAL_SIZE": "2", "OMPI_COMM_WORLD_LOCAL_RANK": "1"},
                  {"SLURM_NTASKS_PER_NODE": "2(x4)", "SLURM_LOCALID": "1"}):
        env = dict(os.environ, WORLD_SIZE="8", **extra)
        for k in ("LOCAL_WORLD_SIZE", "LOCAL_RANK", "DVT_NO_AFFINITY"):
            env.pop(k, None)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout
        info = json.loads(out.strip().splitlines()[-1])
        assert info["pinned"] and info["local_rank"] == 1 and set(info["affinity"]) == seen[1], (extra, info)
    # ONE process inside a multi-task allocation (sbatch --ntasks-per-node=8: the batch step exports the
    # allocation's SLURM_NTASKS_PER_NODE / SLURM_LOCALID=0) is not a rank of anything: it keeps the whole host ...
    env = dict(os.environ, SLURM_NTASKS_PER_NODE="8", SLURM_LOCALID="0", SLURM_NTASKS="8", SLURM_PROCID="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_WORLD_SIZE", "LOCAL_RANK", "DVT_NO_AFFINITY", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout
    info = json.loads(out.strip().splitlines()[-1])
    assert not info["pinned"] and len(info["affinity"]) == n_cpu, info
    # ... and so does one rank per node of a multi-node job (WORLD_SIZE = 2 <= the GPUs of this host, no local variables)
    env = dict(os.environ, WORLD_SIZE="2", RANK="1")
    for k in ("LOCAL_WORLD_SIZE", "LOCAL_RANK", "DVT_NO_AFFINITY", "OMPI_COMM_WORLD_LOCAL_SIZE", "SLURM_NTASKS_PER_NODE",
              "MPI_LOCALNRANKS"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout
    info = json.loads(out.strip().splitlines()[-1])
    assert not info["pinned"] and len(info["affinity"]) == n_cpu, info


def _bench_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE",
                                                             "MASTER_ADDR", "MASTER_PORT")}


def test_bench_plain_command_refuses_only_for_missing_devices():
    """`python bench.py --gpus N` is what the round-end driver runs.  Started plain (no torch.distributed.run
    environment) with N > 1 it must launch N ranks itself; the only refusal is fewer than N visible devices -- on this CPU-only
    container that is the case, and the message says so (not an assertion about torchrun).  No JSON line, rc != 0."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_bench_env(), capture_output=True, text=True)
    assert r.returncode != 0
    assert "2 GPUs requested, 0 visible" in r.stderr, r.stderr[-400:]
    assert "AssertionError" not in r.stderr and '"metric"' not in r.stdout


def test_bench_self_launches_two_ranks_over_gloo():
    """The self-launch path itself, on the CPU stub: `python bench.py --gpus 2 --rendezvous-only` re-launches itself under
    torch.distributed.run with two ranks, both join the process group (gloo here, RCCL on a node), pass the bench's own timed
    bracket (barrier + MAX over ranks + the one gather) and rank 0 prints ONE JSON line naming the group's size."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       env=_bench_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-800:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["rendezvous_only"] and d["n_gpus"] == 2 and d["dist_world_size"] == 2 and d["dist_backend"] == "gloo"
    assert [p["rank"] for p in d["per_rank"]] == [0, 1] and all(p["units"] == 1 for p in d["per_rank"])


def test_bench_rank_checks_its_world_size():
    """Inside a job the ranks still check `--gpus` against the group: a rank of a 1-rank job asked for --gpus 2 stops."""
    import subprocess
    import sys
    env = dict(_bench_env(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr
