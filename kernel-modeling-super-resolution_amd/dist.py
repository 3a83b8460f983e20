"""One-process-per-GPU plumbing over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU).

The hot path shards by independent patches: no data-path collective for degradation / SR inference /
kernel estimation -- only a barrier around timed regions and a MAX of per-rank wall times.  SR *training* adds exactly
one exchange per step: the gradient all-reduce, done here as ONE flat fp32 bucket (the 1.52 M-parameter SR net is a
6 MB message: latency-bound, so a single collective beats per-tensor calls on point-to-point xGMI links).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  A single process (WORLD_SIZE unset or 1) needs no group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    return rank, world, local


def shard_indices(n, rank, world):
    """Static round-robin shard of n independent units: rank r owns r, r+world, ...  (SURVEY 8e)."""
    return list(range(rank, n, world))


def rank_seed(base_seed, rank=None):
    """Seed of THIS rank's counter-based noise stream.  The in-kernel Gaussian noise is addressed by (seed, offset, local
    plane, row, col); with one seed for all ranks sample n of every rank would receive a bit-identical eps field each step,
    unlike the reference's independent torch.randn_like draws (train_gemini.py:137).  Rank r therefore keys its stream with
    base_seed + r * 0x9E3779B97F4A7C15 (mod 2^64, odd multiplier -> distinct keys for distinct ranks); rank 0 (and a
    single process) keeps base_seed, so single-GPU results do not depend on whether a process group exists."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    return (int(base_seed) + int(rank) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX-reduce a python float over all ranks (timed-region wall clock)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


class GradBucket:
    """Flat fp32 gradient bucket for data-parallel training: grads are views into one contiguous buffer, so the
    step's exchange is a single all-reduce (sum) followed by a scale by 1/world."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else "cpu"
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self, async_op=False):
        """Average gradients over ranks.  async_op=True returns the work handle (overlap with the rest of backward);
        call .finish(handle) before the optimizer step."""
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return None
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        if not async_op:
            self.flat.div_(dist.get_world_size())
        return work

    def finish(self, work):
        if work is not None:
            work.wait()
            self.flat.div_(dist.get_world_size())


def broadcast_parameters(module, src=0):
    """Make every rank start from rank `src`'s weights."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)
