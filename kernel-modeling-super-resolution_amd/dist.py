"""One-process-per-GPU plumbing over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU).

The hot path shards by independent patches: no data-path collective for degradation / SR inference /
kernel estimation -- only a barrier around timed regions and a MAX of per-rank wall times.  SR *training* adds exactly
one exchange per step: the gradient all-reduce, done here as ONE flat fp32 bucket (the 1.52 M-parameter SR net is a
6 MB message: latency-bound, so a single collective beats per-tensor calls on point-to-point xGMI links).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force=False):
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  A single process (WORLD_SIZE unset or 1) needs no group; force=True creates the
    (legal) one-rank group anyway -- the RCCL smoke test on a one-GPU box."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (read when the HSA runtime starts): RCCL needs it on this pool
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


def sample_base(local_batch, rank=None):
    """Global index of THIS rank's first sample when every rank holds a contiguous block of `local_batch` samples of the
    step's global batch (rank r owns samples r*B .. r*B+B-1).  The in-kernel Gaussian noise is addressed by (seed, offset,
    GLOBAL sample, band, row, col) (kmsr_degrade_desc.noise_sample_base / _step), so with this base the ranks draw independent
    eps fields (the reference's torch.randn_like per sample, train_gemini.py:137) AND the concatenated result equals what one
    process draws for the whole batch: results do not depend on the world size.  For the round-robin shard
    patch_ids[rank::world] use sample_base=rank, sample_step=world instead."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    return int(rank) * int(local_batch)


def batch_offsets(local_batch, device="cpu"):
    """(global index of this rank's first sample, global batch) from ONE all_gather of the ranks' local batch sizes
    (exclusive prefix sum).  The form for ragged shards (a short last batch, ranks with different shard sizes), where
    rank * local_batch would overlap or leave gaps in the global sample addressing; collective, so every rank calls it."""
    if not dist.is_initialized():
        return 0, int(local_batch)
    t = torch.tensor([int(local_batch)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(sizes, t)
    sizes = [int(v.item()) for v in sizes]
    return sum(sizes[:dist.get_rank()]), sum(sizes)


def assert_equal_batch(local_batch, device="cpu"):
    """One-time guard for callers that address noise with sample_base(local_batch) = rank * local_batch: that is only a
    partition of the global batch when every rank holds the same number of samples.  Raises on every rank otherwise."""
    if not dist.is_initialized():
        return
    t = torch.tensor([int(local_batch), -int(local_batch)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    lo, hi = -int(t[1].item()), int(t[0].item())
    if lo != hi:
        raise ValueError(f"data-parallel ranks hold different batch sizes ({lo} .. {hi}): rank * local_batch does not address the "
                         "global samples; pass explicit offsets (dist.batch_offsets) as the module's `sample_addr`")


def ranks_agree(t):
    """(ranks seen, all equal) for a tensor that must be bit-identical on every rank (the averaged flat gradient after the
    step's all-reduce; broadcast weights): one all_gather of (rank, int64 sum of the raw 32-bit words, int64 sum of the words
    weighted by position mod 8191).  bench.py prints `ranks seen` so the driver's multi-GPU record proves N ranks met."""
    if not dist.is_initialized():
        return 1, True
    w = t.detach().contiguous().view(-1).view(torch.int32).to(torch.int64)
    pos = torch.arange(w.numel(), device=w.device, dtype=torch.int64) % 8191 + 1
    mine = torch.stack([torch.tensor(dist.get_rank(), device=w.device, dtype=torch.int64), w.sum(), (w * pos).sum()])
    got = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(got, mine)
    got = torch.stack(got).cpu()
    seen = len(set(int(r) for r in got[:, 0]))
    return seen, bool((got[:, 1:] == got[0, 1:]).all())


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
    """Flat fp32 gradient bucket for data-parallel training: grads are views into one contiguous buffer and the step's
    exchange is an all-reduce (sum) of that buffer followed by a scale by 1/world.

    Overlap with backward: the buffer is cut into `chunks` slices on parameter boundaries.  Backward produces gradients in
    (roughly) reverse parameter order, so the LAST slice completes first; a post-accumulate hook counts a slice's parameters
    and launches its asynchronous all-reduce the moment the slice is complete (RCCL runs it on its own stream behind the
    kernels already queued), while backward goes on with the earlier layers.  finish() launches whatever is left (the first
    slice; slices with unused parameters), waits and scales.  With a 6 MB gradient two slices are enough: each message stays
    large enough to be bandwidth- rather than latency-bound on the point-to-point xGMI links.

    Contract of the overlap: between zero(backwards=n) and finish() exactly n backward passes accumulate into the bucket (n = 1
    unless the step uses gradient-accumulation micro-steps or back-propagates several losses separately); a slice is launched
    when its parameters have received their n-th gradient.  A hook that fires for a slice already in flight (more backward
    passes than declared) raises instead of silently adding local gradients to a reduced slice.  `always_reduce=True` runs
    the collective at world size 1 as well (the RCCL smoke test: the sum over one rank is the identity).

    Round 4 -- no accumulate kernels: zero() sets the gradients to None, so autograd's first (usually only) gradient of a parameter is taken
    over as produced instead of being added into a zeroed view (76 adds + the memset per SR training step); a slice is GATHERED into the
    flat buffer -- one multi-tensor copy -- when it completes (from the hook when the overlap is armed, else in all_reduce_mean() / finish()),
    and from then on `p.grad` is the view of the flat buffer again: what the optimizer, the collective and `flat` readers see is unchanged."""

    def __init__(self, params, chunks=2, always_reduce=False):
        self.always_reduce = bool(always_reduce)
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else "cpu"
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off, offs, self._views = 0, [], []
        for p in self.params:
            self._views.append(self.flat[off:off + p.numel()].view_as(p))
            p.grad = self._views[-1]
            offs.append(off)
            off += p.numel()
        # slice boundaries: parameter boundaries closest to k*n/chunks
        chunks = max(1, min(int(chunks), len(self.params) or 1))
        cuts = [0]
        for k in range(1, chunks):
            tgt = n * k // chunks
            b = min(range(len(offs)), key=lambda i: abs(offs[i] - tgt)) if offs else 0
            if b > cuts[-1]:
                cuts.append(b)
        cuts.append(len(self.params))
        self._slices = []                              # (first param, end param, flat start, flat end)
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b > a:
                self._slices.append((a, b, offs[a], offs[b] if b < len(offs) else n))
        self._slice_of = {}
        for k, (a, b, _, _) in enumerate(self._slices):
            for p in self.params[a:b]:
                self._slice_of[id(p)] = k
        self._armed = False
        self._pending, self._works = [], []
        self._gathered = [True] * len(self._slices)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _distributed(self):
        return dist.is_initialized() and (dist.get_world_size() > 1 or self.always_reduce)

    def _gather(self, k):
        """Slice k's gradients -> the flat buffer (one multi-tensor copy); the parameters' .grad become the buffer's views again."""
        if self._gathered[k]:
            return
        a, b, _, _ = self._slices[k]
        dst, src = [], []
        for p, v in zip(self.params[a:b], self._views[a:b]):
            g = p.grad
            if g is None:
                v.zero_()                               # no gradient reached this parameter in this step
            elif g.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(g.detach())
            p.grad = v
        if dst:
            with torch.no_grad():
                if hasattr(torch, "_foreach_copy_"):
                    torch._foreach_copy_(dst, src)          # one multi-tensor launch
                else:
                    for v, g in zip(dst, src):
                        v.copy_(g)
        self._gathered[k] = True

    def _launch(self, k):
        self._gather(k)
        _, _, a, b = self._slices[k]
        self._works[k] = dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=True)

    def _on_grad(self, p):
        if not self._armed:
            return
        k = self._slice_of[id(p)]
        if self._works[k] is not None:
            self._armed = False
            raise RuntimeError("GradBucket: a gradient arrived for a slice whose all-reduce is already in flight -- more backward "
                               "passes than zero(backwards=n) declared; the averaged gradients would be wrong")
        self._pending[k] -= 1
        if self._pending[k] == 0:
            self._launch(k)

    def zero(self, backwards=1, overlap=True):
        """Drop the gradients (None: the next backward's gradients are taken over, not added to zeros) and arm the overlap hooks for the `backwards` backward passes that follow (overlap=False: no
        hook launches anything, finish() / all_reduce_mean() reduce the whole bucket)."""
        for p in self.params:
            p.grad = None
        self._gathered = [False] * len(self._slices)
        self._pending = [(b - a) * max(int(backwards), 1) for a, b, _, _ in self._slices]
        self._works = [None] * len(self._slices)
        self._armed = bool(overlap) and self._distributed()

    def launched(self):
        """Slices whose all-reduce has already been launched (by the hooks, during backward)."""
        return sum(w is not None for w in self._works)

    def all_reduce_mean(self, async_op=False):
        """Average gradients over ranks.  async_op=True returns the work handles (call .finish(handles) before the optimizer
        step); slices the hooks already launched during backward are not launched again."""
        self._armed = False
        for k in range(len(self._slices)):
            self._gather(k)
        if not self._distributed():
            return None
        if len(self._works) != len(self._slices):
            self._works = [None] * len(self._slices)
        for k in range(len(self._slices)):
            if self._works[k] is None:
                self._launch(k)
        works, self._works = self._works, [None] * len(self._slices)
        if async_op:
            return works
        self.finish(works)
        return None

    def finish(self, work=None):
        if work is None:
            work = self.all_reduce_mean(async_op=True)
        if not self._distributed() or work is None:
            return
        for w in work:
            w.wait()
        self.flat.div_(dist.get_world_size())


def broadcast_parameters(module, src=0):
    """Make every rank start from rank `src`'s weights."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)
