"""Process-group plumbing of the N > 1 runs (one process per GPU, `torch.distributed`).

Stage 1 shards images statically (`misc.shard_range`, the arithmetic of the reference's
sample_scripts/stage1.sh:8-20: eight independent processes, disjoint `--start_idx/--num_imgs`
slices) and needs NO collective in its data path.  What the ranks do share is bookkeeping: a
barrier on either side of a timed region, the maximum elapsed time over ranks, and ONE gather of
per-rank (images, seconds) at the end of a run.  Backend "nccl" is RCCL on ROCm; on a CPU-only
host (the world-size-2 tests) the same code runs over gloo.
"""
from __future__ import annotations

import os
import time
from typing import Callable

import torch
import torch.distributed as dist


def env_ranks() -> tuple[int, int, int]:
    """(rank, world_size, local_rank) as exported by torch.distributed.run; (0, 1, 0) standalone."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)),
            int(os.environ.get("LOCAL_RANK", 0)))


def _first_env(names, default):
    """Value of the first variable of `names` that is set to something int()-able (SLURM writes e.g. "8(x2)": leading digits)."""
    for name in names:
        v = os.environ.get(name, "")
        digits = 0
        while digits < len(v) and v[digits].isdigit():
            digits += 1
        if digits:
            return v[:digits]
    return default


_warned: set = set()


def _warn_once(msg: str) -> None:
    if msg not in _warned:
        _warned.add(msg)
        print(msg, flush=True)


_pin_info: dict | None = None        # what pin_host_threads did for this process (it acts once)
_orig_affinity: list | None = None   # the CPU set the process started with


def pin_host_threads(local_rank: int | None = None, local_world: int | None = None) -> dict:
    """Host budget of one rank on a multi-GPU node.  Every rank runs ~4 host threads (extractor,
    fit, retire + write, index-stream look-ahead) that spend most of their life blocked on a full HIP queue, plus
    the numpy index draws (2 M draws = ~20 ms per image) and torch's intra-op pool.  Eight ranks on one host must not
    fight over the same cores: each rank gets a contiguous slice of the CPUs this process may run on
    (`sched_setaffinity`) and caps its torch / OpenMP pools to that slice (at most 8 threads).  A single-process run
    (local_world == 1) is left alone.  DVT_NO_AFFINITY=1 disables the pinning; returns what was done."""
    global _pin_info, _orig_affinity
    if local_rank is None:
        local_rank = int(_first_env(("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID", "MPI_LOCALRANKID"), 0))
    # the ranks of THIS host only.  torch.distributed.run exports LOCAL_WORLD_SIZE; mpirun / srun / hand-rolled spawns
    # export their own names (with only LOCAL_WORLD_SIZE read, such launches silently lost the pinning and
    # eight ranks fought over the same cores again).  Last resort: WORLD_SIZE when it cannot span more than this host's GPUs.
    if local_world is None:
        # The scheduler's per-node variables describe the ALLOCATION, not this process: a single `python bench.py` inside an
        # sbatch allocation made with --ntasks-per-node=8 sees SLURM_NTASKS_PER_NODE=8 / SLURM_LOCALID=0 and must keep the whole
        # host.  They count only when this process really is one rank of several: a world size > 1 exported by the
        # LAUNCHER of this process (torch.distributed.run's / a wrapper's WORLD_SIZE -- which dist.init's env:// rendezvous needs
        # anyway --, mpirun's OMPI_COMM_WORLD_SIZE, PMI_SIZE); SLURM's allocation-level variables alone never suffice.
        world = int(_first_env(("WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE"), 1))
        is_rank = world > 1
        local_world = int(_first_env(("LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "SLURM_NTASKS_PER_NODE",
                                      "MPI_LOCALNRANKS"), 0)) if is_rank else 0
        if local_world <= 0 and is_rank:
            n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
            # last resort: WORLD_SIZE is this host's rank count only when the launcher ALSO says the ranks are local
            # (LOCAL_RANK set: torch.distributed.run single-node) -- one rank per node of a multi-node job has WORLD_SIZE > 1
            # and no local rank, and keeps its host
            if 1 < world <= n_dev and "LOCAL_RANK" in os.environ:
                local_world = world
            elif world > 1:
                _warn_once(f"dvt_amd.dist: WORLD_SIZE={world} but no local world size in the environment (LOCAL_WORLD_SIZE / "
                           "OMPI_COMM_WORLD_LOCAL_SIZE / SLURM_NTASKS_PER_NODE / MPI_LOCALNRANKS): host threads are NOT pinned")
    info = {"local_rank": local_rank, "local_world": local_world, "pinned": False}
    if local_world <= 1 or os.environ.get("DVT_NO_AFFINITY") == "1" or not hasattr(os, "sched_setaffinity"):
        return info
    if _pin_info is not None and _pin_info.get("local_rank") == local_rank and _pin_info.get("local_world") == local_world:
        return _pin_info  # once per process: a second init() (stage1.main, then stage2.train) must not re-slice the slice
    if _orig_affinity is None:
        _orig_affinity = sorted(os.sched_getaffinity(0))
    cpus = _orig_affinity  # always slice the set the process STARTED with
    per = max(1, len(cpus) // local_world)
    mine = cpus[local_rank * per:(local_rank + 1) * per] or cpus
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return info
    n = max(1, min(8, len(mine)))
    torch.set_num_threads(n)  # (the OpenMP pool already exists once torch is imported: the env variable would be a no-op)
    info.update(pinned=True, cpus=[mine[0], mine[-1]], n_cpus=len(mine), torch_threads=n)
    _pin_info = info
    return info


def init(device: torch.device, world: int) -> bool:
    """Join the env:// rendezvous when world > 1 (RCCL for a HIP device, gloo for cpu)."""
    if world <= 1 or dist.is_initialized():
        return dist.is_initialized()
    pin_host_threads()
    if device.type == "cuda":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group("gloo")
    return True


def _sync(device: torch.device) -> None:
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def barrier(device: torch.device) -> None:
    """All queued device work done on every rank (device sync, process barrier, device sync)."""
    _sync(device)
    if dist.is_initialized():
        dist.barrier()
    _sync(device)


def gather_stats(values, device: torch.device) -> list[list[float]]:
    """The one collective of a run: every rank's small vector of doubles, on every rank."""
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if not dist.is_initialized():
        return [t.tolist()]
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]


def timed(fn: Callable[[], int], device: torch.device) -> tuple[int, float, list[list[float]]]:
    """Run `fn` (returns the units this rank processed) between two barriers.
    Returns (units of this rank, MAX elapsed seconds over ranks, per-rank [units, seconds])."""
    barrier(device)
    t0 = time.perf_counter()
    n = int(fn())
    _sync(device)
    mine = time.perf_counter() - t0
    barrier(device)
    elapsed = time.perf_counter() - t0
    per_rank = gather_stats([n, mine, elapsed], device)
    return n, max(r[2] for r in per_rank), [[r[0], r[1]] for r in per_rank]


def finish() -> None:
    if dist.is_initialized():
        dist.destroy_process_group()
