"""One process per GPU; recordings shard across ranks; RCCL (torch.distributed backend "nccl" on ROCm) is used once,
for the final WER gather.  The reference has no distributed code at all: it pins one process per GPU
with CUDA_VISIBLE_DEVICES (every file in reference lcasr/launch_scripts/) — recordings are independent because
eval_fn restores the weights and builds a fresh optimiser per call (reference lcasr/lib.py:494,636-637).

No gradient or activation collective exists or should exist: adaptation inside one recording is strictly sequential."""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def local_device_index(local_rank):
    """GPU of this rank: LOCAL_RANK, folded onto the visible devices when ranks outnumber them (rehearsal only)."""
    n = torch.cuda.device_count()
    return local_rank % n if n > 0 else 0


def init(backend=None, force=False):
    """Initialise the default process group from the torchrun environment; returns (rank, local_rank, world).
    A single process needs no group (every helper below then returns its local value); `force=True` creates the one-rank group
    anyway, so that the collectives really go through the backend (tests/test_dist_gpu.py: RCCL on the box's one GPU)."""
    rank, local_rank, world = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            # DYN_DIST_BACKEND=gloo: rehearsal of the N > 1 code path on a box with fewer GPUs than ranks
            backend = os.environ.get("DYN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_device_index(local_rank))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shutdown():
    """Destroy the default group (a process that initialised RCCL should tear it down before it exits)."""
    if dist.is_initialized():
        dist.destroy_process_group()


def shard_longest_first(lengths, world):
    """LPT bin packing: recordings sorted by length (longest first), each to the currently lightest rank.
    Returns a list (per rank) of recording indices; deterministic on every rank."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    load = [0] * world
    bins = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        bins[r].append(i)
        load[r] += lengths[i]
    return [sorted(b) for b in bins]


def _device_for(backend_tensor_device=None):
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def all_reduce_counts(counts):
    """Sum the 4 int64 edit counters (ins, del, sub, words) over ranks — the one collective of the path."""
    if not dist.is_initialized():
        return tuple(int(c) for c in counts)
    t = torch.tensor(list(counts), dtype=torch.int64, device=_device_for())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return tuple(int(x) for x in t.cpu().tolist())


def gather_records(records):
    """Gather per-recording result dicts (id, hypothesis, gold, elapsed) to every rank, ordered by recording index."""
    if not dist.is_initialized():
        return sorted(records, key=lambda r: r["index"])
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, records)
    flat = [r for part in out for r in part]
    return sorted(flat, key=lambda r: r["index"])


def max_over_ranks(value):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device_for())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])   # RCCL: name the device instead of letting torch guess it
        else:
            dist.barrier()
