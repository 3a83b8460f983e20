"""world_size-2 gloo tests (CPU) of the multi-process plumbing used by bench.py --gpus N and by data-parallel training."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from kmsr_amd import dist as kd
    r, w, _ = kd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    # (1) shards are disjoint and complete
    mine = kd.shard_indices(11, r, w)
    gathered = [None] * w
    dist.all_gather_object(gathered, mine)
    flat = sorted(i for g in gathered for i in g)
    assert flat == list(range(11))
    # (2) timing reductions
    assert kd.max_over_ranks(1.0 + rank) == float(world)
    assert kd.sum_over_ranks(1.0) == float(world)
    # (3) data-parallel gradients: one flat bucket, mean over ranks == gradient of the mean loss over all samples
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 3, 3, padding=1))
    if rank != 0:
        for p_ in net.parameters():
            p_.data.add_(1.0)                      # deliberately different start
    kd.broadcast_parameters(net, src=0)
    bucket = kd.GradBucket(net.parameters())
    g = torch.Generator().manual_seed(42)
    x_all, y_all = torch.randn(4, 3, 8, 8, generator=g), torch.randn(4, 3, 8, 8, generator=g)
    sl = slice(rank * 2, rank * 2 + 2)            # 2 samples per rank
    bucket.zero()
    torch.nn.functional.mse_loss(net(x_all[sl]), y_all[sl]).backward()
    work = bucket.all_reduce_mean(async_op=True)
    bucket.finish(work)
    ref = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 3, 3, padding=1))
    ref.load_state_dict(net.state_dict())
    torch.nn.functional.mse_loss(ref(x_all), y_all).backward()
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-7)
    # (3b) overlap: with hooks armed by zero(), the upper slice of the flat gradient is already in flight when backward returns
    bucket.zero()
    torch.nn.functional.mse_loss(net(x_all[sl]), y_all[sl]).backward()
    assert len(bucket._slices) == 2 and bucket.launched() >= 1, (bucket._slices, bucket.launched())
    bucket.finish()
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-7)
    assert kd.ranks_agree(bucket.flat) == (world, True)            # averaged gradients are bit-identical on all ranks
    assert kd.ranks_agree(bucket.flat + rank) == (world, False)
    # (3c) two backward passes into one bucket (gradient-accumulation micro-steps): declared with zero(backwards=2), the slices are
    #      launched only after their parameters' SECOND gradient; the mean equals the single-process sum of both micro-batch losses
    bucket.zero(backwards=2)
    torch.nn.functional.mse_loss(net(x_all[sl][:1]), y_all[sl][:1]).backward()
    assert bucket.launched() == 0
    torch.nn.functional.mse_loss(net(x_all[sl][1:]), y_all[sl][1:]).backward()
    assert bucket.launched() >= 1
    bucket.finish()
    ref.zero_grad()
    for r_ in range(world):
        for m_ in range(2):
            i_ = r_ * 2 + m_
            (torch.nn.functional.mse_loss(ref(x_all[i_:i_ + 1]), y_all[i_:i_ + 1]) / world).backward()
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-7)
    # (3d) an UNDECLARED second backward must not silently add local gradients to a slice that is already reduced / in flight
    bucket.zero()
    torch.nn.functional.mse_loss(net(x_all[sl]), y_all[sl]).backward()
    with pytest.raises(RuntimeError, match="already in flight"):
        torch.nn.functional.mse_loss(net(x_all[sl]), y_all[sl]).backward()
    bucket.finish()                                  # drains what was launched; the bucket is reusable afterwards
    bucket.zero(overlap=False)                       # opt-out: nothing is launched from the hooks
    torch.nn.functional.mse_loss(net(x_all[sl]), y_all[sl]).backward()
    torch.nn.functional.mse_loss(net(x_all[sl]), y_all[sl]).backward()
    assert bucket.launched() == 0
    bucket.finish()
    # (3f) round 4: gradients are taken over from autograd and gathered into the flat buffer per slice.  A parameter no gradient reaches
    #      (unused branch) must read as zeros in the buffer, p.grad must be the buffer's view again after finish(), and stale values of
    #      the previous step must not survive
    class _Two(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.unused, self.b = torch.nn.Linear(3, 3), torch.nn.Linear(3, 3), torch.nn.Linear(3, 2)

        def forward(self, x):
            return self.b(torch.relu(self.a(x)))
    torch.manual_seed(1)
    two = _Two()
    kd.broadcast_parameters(two, src=0)
    b2 = kd.GradBucket(two.parameters())
    xs = torch.randn(4, 3, generator=torch.Generator().manual_seed(5))
    for rep in range(2):
        b2.flat.fill_(123.0)                        # stale content
        b2.zero()
        assert all(p_.grad is None for p_ in two.parameters())
        two(xs[rank * 2:rank * 2 + 2]).square().mean().backward()
        b2.finish()
        for p_, v_ in zip(b2.params, b2._views):
            assert p_.grad.data_ptr() == v_.data_ptr()
        assert float(two.unused.weight.grad.abs().max()) == 0.0 and float(two.unused.bias.grad.abs().max()) == 0.0
        assert kd.ranks_agree(b2.flat) == (world, True)
    ref2 = _Two()
    ref2.load_state_dict(two.state_dict())
    ref2(xs).square().mean().backward()
    for a, b in zip(two.parameters(), ref2.parameters()):
        if b.grad is not None:
            assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-7)
    # (3e) ragged shards: explicit global sample offsets from one all_gather; the equal-batch guard trips on every rank
    assert kd.batch_offsets(3 + rank) == (0 if rank == 0 else 3, 7)
    kd.assert_equal_batch(4)
    with pytest.raises(ValueError, match="different batch sizes"):
        kd.assert_equal_batch(4 + rank)
    # (4) world-size-invariant noise: the counter-based Gaussian stream is addressed by the GLOBAL sample index
    #     (kd.sample_base = rank * local batch), so the ranks' fields differ AND their concatenation is bit for bit the field a
    #     single process draws for the whole batch (round-2 advisor finding)
    import numpy as np
    import oracle
    Bl, Cc = 3, 5
    base = kd.sample_base(Bl)
    assert base == rank * Bl and kd.sample_base(Bl, rank=0) == 0
    eps = oracle.gaussian_eps(Bl * Cc, 8, 8, seed=42, offset=7, plane_base=base * Cc)
    fields = [None] * w
    dist.all_gather_object(fields, eps)
    assert not np.array_equal(fields[0], fields[1]) and abs(float(np.corrcoef(fields[0].ravel(), fields[1].ravel())[0, 1])) < 0.3
    assert np.array_equal(np.concatenate(fields, axis=0), oracle.gaussian_eps(w * Bl * Cc, 8, 8, seed=42, offset=7))
    # grads are views of the flat buffer (a single collective per step)
    assert all(p_.grad.data_ptr() >= bucket.flat.data_ptr() for p_ in net.parameters())
    kd.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, "ok"), (1, "ok")]


def test_single_process_is_a_noop():
    sys.path.insert(0, ROOT)
    from kmsr_amd import dist as kd
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    assert kd.init_from_env() == (0, 1, 0)
    assert kd.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    assert kd.max_over_ranks(3.5) == 3.5
    b = kd.GradBucket(torch.nn.Linear(2, 2).parameters())
    assert b.all_reduce_mean() is None
