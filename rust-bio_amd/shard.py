"""Multi-GPU sharding of independent units (pairs / queries): one process per GPU, contiguous
ranges, no exchange during compute, and ONE all-gather of fixed-size result records at the end
(north_star).  Backend "nccl" is RCCL over xGMI on ROCm; the same code runs on
"gloo" for the CPU tests."""
import os

import torch
import torch.distributed as dist


def env_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def single_gpu_mode():
    """Several ranks on ONE GPU (tests on a 1-GPU box, BENCH_SINGLE_GPU=1): every rank uses device 0 and the
    collective runs over gloo on host copies of the records — RCCL needs one device per rank."""
    if os.environ.get("BENCH_SINGLE_GPU") == "1":
        return True
    return torch.cuda.is_available() and env_world()[2] > 1 and torch.cuda.device_count() == 1


def device_index(local_rank):
    return 0 if single_gpu_mode() else local_rank


def init_process_group(backend=None):
    """Returns (rank, local device index, world)."""
    rank, local_rank, world = env_world()
    dev = device_index(local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() and not single_gpu_mode() else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", dev)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, dev, world


def _host_side():
    """gloo carries host tensors: device records take a round trip through host memory"""
    return dist.get_backend() == "gloo"


def broadcast(t, src=0):
    """Broadcast of a replicated table (the BWT, suffix-array samples) from rank `src`, in place."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t
    if _host_side() and t.is_cuda:
        h = t.cpu()
        dist.broadcast(h, src)
        t.copy_(h)
    else:
        dist.broadcast(t, src)
    return t


def partition(n_units, rank, world):
    """Contiguous range [lo, hi) of rank `rank`: [rank*N/W, (rank+1)*N/W)."""
    return n_units * rank // world, n_units * (rank + 1) // world


def partition_balanced(costs, world):
    """Contiguous ranges balanced by a per-unit cost (sum of cells / pattern lengths);
    returns world+1 boundaries."""
    c = torch.cumsum(torch.as_tensor(costs, dtype=torch.float64), 0)
    total = float(c[-1]) if len(c) else 0.0
    bounds = [0]
    for r in range(1, world):
        bounds.append(int(torch.searchsorted(c, torch.tensor(total * r / world, dtype=torch.float64))))
    bounds.append(len(c))
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


def gather_records(local, counts=None):
    """The single collective: all-gather of this rank's fixed-size result records
    (tensor [n_local, k]).  With equal shard sizes one all_gather_into_tensor; ragged shards
    are padded to the largest and trimmed afterwards."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    if _host_side() and local.is_cuda:
        return gather_records(local.cpu(), counts).to(local.device)
    world = dist.get_world_size()
    if counts is None:
        cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        allc = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(allc, cnt)
        counts = [int(c.item()) for c in allc]
    mx = max(counts)
    if all(c == mx for c in counts):
        out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * mx:r * mx + counts[r]] for r in range(world)])


def max_over_ranks(seconds, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if _host_side() else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
