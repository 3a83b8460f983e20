"""GPU parity tests of the fused degradation path, called through the C ABI (kmsr_degrade_f32).

Three layers of evidence per case:
  1. golden vectors produced by the reference itself (tests/golden/*.npz)            -- rtol 2e-5
  2. the CPU oracle on the same seeded inputs (oracle/, pinned in test_oracle_golden) -- rtol 2e-5
  3. the row-streaming fast kernel against the generic kernel                        -- BIT-EXACT
     (same tap table, same (a,b) fmaf order), which also proves the output index map is identical.
Tolerance: north_star asks for fp32 within 1e-4 rel of the reference CPU path; the tests hold 2e-5.
Full BASELINE sizes are covered through size-independent properties (linearity, constants, impulse
response, fast==generic) because the oracle would take minutes there.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import kmsr_amd
import oracle
from conftest import assert_close, golden, landsat_like
from kmsr_amd import _lib as L

pytestmark = pytest.mark.gpu
RTOL = 2e-5

VA = golden("va_cases.npz")
VB = golden("vb_cases.npz")
K = golden("kernels.npz")
VA_NAMES = sorted({k.split("__")[0] for k in VA.files if "__" in k and not k.startswith("nan_case")})
# cases whose geometry must be served by the fast (row-streaming) kernel
VA_FAST = {"c5_512_f4_moe0", "c5_512_f8_moe0", "c5_256_f8_kgan", "c3_256_f4_merged2d", "c5_96_f6_is_f4", "c5_64_f2",
           "c5_64_f4_zero_band", "c5_64_f4_unnormalised"}


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def _va(img, kernel, f, **kw):
    k = torch.as_tensor(kernel)
    if k.ndim == 2:
        k = k.unsqueeze(0).repeat(img.shape[0], 1, 1)
    return kmsr_amd.degrade_batch(dev(img)[None], k.cuda()[None], f, down="avgpool", pad="replicate", normalize=True,
                                  **kw)[0].cpu().numpy()


@pytest.mark.parametrize("name", VA_NAMES)
def test_va_golden_and_oracle(name, gpu_device):
    """apply_kernel_degradation == reference (C_30:68-124) on every golden case, through the drop-in signature."""
    Cc, H, W, f, seed = (int(v) for v in VA[name + "__meta"])
    img = landsat_like((Cc, H, W), seed)
    kern = VA[name + "__kernel"]
    out = kmsr_amd.apply_kernel_degradation(torch.from_numpy(img), torch.from_numpy(kern), f)
    assert out.device.type == "cpu" and out.dtype == torch.float32          # CPU in -> CPU out like the reference
    assert_close(out.numpy(), VA[name + "__out"], rtol=RTOL, what=f"{name} vs golden")
    assert_close(out.numpy(), oracle.apply_kernel_degradation(img, kern, f), rtol=RTOL, what=f"{name} vs oracle")
    k3 = torch.from_numpy(kern) if kern.ndim == 3 else torch.from_numpy(kern)[None].repeat(Cc, 1, 1)
    fast = kmsr_amd.uses_fast_path(dev(img)[None], k3[None], f, "avgpool")
    assert fast == (name in VA_FAST), f"{name}: fast path = {fast}"
    # fast kernel vs generic kernel: bit-exact
    gen = _va(img, kern, f, flags=L.FLAG_FORCE_GENERIC)
    assert np.array_equal(out.numpy(), gen), f"{name}: fast and generic kernels differ"


def test_va_gpu_tensor_stays_on_gpu(gpu_device):
    img = dev(landsat_like((5, 64, 64), 9))
    out = kmsr_amd.apply_kernel_degradation(img, dev(K["moe_kernels"][0]), 4)
    assert out.is_cuda and tuple(out.shape) == (5, 16, 16)


def test_va_nan_propagation(gpu_device):
    out = _va(VA["nan_case__in"], VA["nan_case__kernel"], 4)
    assert_close(out, VA["nan_case__out"], rtol=RTOL, what="nan case")
    assert np.isnan(out).sum() == 16 and not np.isnan(out[[0, 2, 3, 4]]).any()
    gen = _va(VA["nan_case__in"], VA["nan_case__kernel"], 4, flags=L.FLAG_FORCE_GENERIC)
    assert np.array_equal(np.isnan(out), np.isnan(gen))


def test_va_error_behaviour(gpu_device):
    img = torch.zeros(3, 32, 32)
    with pytest.raises(AssertionError):                    # C_30:88
        kmsr_amd.apply_kernel_degradation(img, torch.ones(5, 13, 13), 4)
    with pytest.raises(ValueError):                        # C_31:69
        kmsr_amd.apply_kernel_degradation(img, torch.ones(1, 3, 13, 13), 4)
    with pytest.raises(NotImplementedError):               # even kernel
        kmsr_amd.apply_kernel_degradation(img, torch.ones(3, 12, 12), 4)
    # image smaller than the pooling factor -> empty LR, like the reference's avg_pool cascade on tiny inputs
    out = kmsr_amd.degrade_batch(torch.zeros(1, 3, 4, 4).cuda(), torch.ones(1, 3, 13, 13).cuda(), 8)
    assert tuple(out.shape) == (1, 3, 0, 0)
    # workspace too small -> KMSR_E_NOMEM, kernel_idx missing with K not in {1,N} -> KMSR_E_SHAPE
    lib = L.load()
    hr = torch.zeros(2, 5, 64, 64).cuda()
    ker = torch.ones(3, 5, 13, 13).cuda()
    lr = torch.zeros(2, 5, 16, 16).cuda()
    d = L.DegradeDesc()
    d.N, d.C, d.H, d.W = 2, 5, 64, 64
    d.hr_stride_n, d.hr_stride_c, d.hr_stride_h = hr.stride(0), hr.stride(1), hr.stride(2)
    d.n_kernels, d.kh, d.kw, d.factor = 3, 13, 13, 4
    ws = torch.zeros(16).cuda()
    rc = lib.kmsr_degrade_f32(d, L.ptr(hr), L.ptr(ker), None, None, None, None, L.ptr(lr), L.ptr(ws), 64, None)
    assert rc == L.KMSR_E_SHAPE
    idx = torch.zeros(2, dtype=torch.int32).cuda()
    rc = lib.kmsr_degrade_f32(d, L.ptr(hr), L.ptr(ker), L.ptr(idx), None, None, None, L.ptr(lr), L.ptr(ws), 64, None)
    assert rc == L.KMSR_E_NOMEM and b"workspace" in lib.kmsr_last_error()
    # kernel / noise-pool indices outside their bank: IndexError on the host, never a device read out of bounds
    hr5 = torch.rand(2, 5, 64, 64).cuda()
    bank = torch.rand(3, 5, 13, 13).cuda()
    for bad in (torch.tensor([0, 3]), torch.tensor([-1, 1]), torch.tensor([0, 3]).cuda()):
        with pytest.raises(IndexError):
            kmsr_amd.degrade_batch(hr5, bank, 4, down="stride", pad="zero", normalize=False, kernel_idx=bad)
    pool = torch.rand(4, 5, 16, 16).cuda()
    with pytest.raises(IndexError):
        kmsr_amd.degrade_batch(hr5, bank[:1], 4, down="stride", pad="zero", normalize=False, noise="pool", noise_pool=pool,
                               noise_idx=torch.tensor([1, 4]))
    ok = kmsr_amd.degrade_batch(hr5, bank, 4, down="stride", pad="zero", normalize=False, kernel_idx=torch.tensor([2, 0]).cuda())
    assert tuple(ok.shape) == (2, 5, 16, 16)
    # the "already validated" cache is bound to the tensor OBJECT: a good device index tensor, freed, then a bad one of the same
    # size that the caching allocator puts on the same block (same data_ptr, version 0) must still be range-checked
    good = torch.tensor([1, 2], dtype=torch.int32).cuda()
    kmsr_amd.degrade_batch(hr5, bank, 4, down="stride", pad="zero", normalize=False, kernel_idx=good)
    addr = good.data_ptr()
    del good
    bad = torch.tensor([1, 5], dtype=torch.int32).cuda()
    if bad.data_ptr() != addr:
        print("note: allocator did not recycle the block; the check below still must fire")
    with pytest.raises(IndexError):
        kmsr_amd.degrade_batch(hr5, bank, 4, down="stride", pad="zero", normalize=False, kernel_idx=bad)


@pytest.mark.parametrize("name,fast", [("b4_256", True), ("b3_250x253", False), ("b2_64", True)])
def test_vb_golden_and_oracle(name, fast, gpu_device):
    """train_gemini.py:124-134: zero pad, per-sample kernels by index, [::4, ::4]."""
    B, Cc, H, W, seed = (int(v) for v in VB[name + "__meta"])
    x = landsat_like((B, Cc, H, W), seed)
    kid = VB[name + "__kid"]
    bank = dev(K["moe_kernels"])
    xd = dev(x)
    out = kmsr_amd.degrade_batch(xd, bank, 4, down="stride", pad="zero", normalize=False, kernel_idx=dev(kid))
    assert_close(out.cpu().numpy(), VB[name + "__out"], rtol=RTOL, what=f"{name} vs golden")
    assert_close(out.cpu().numpy(), oracle.vb_blur_stride(x, K["moe_kernels"][kid], 4), rtol=RTOL, what=f"{name} vs oracle")
    assert kmsr_amd.uses_fast_path(xd, bank, 4, "stride") == fast
    gen = kmsr_amd.degrade_batch(xd, bank, 4, down="stride", pad="zero", normalize=False, kernel_idx=dev(kid),
                                 flags=L.FLAG_FORCE_GENERIC)
    assert torch.equal(out, gen)
    # per-sample kernels given directly (K == N, no index) must agree with the indexed form
    direct = kmsr_amd.degrade_batch(xd, bank[torch.as_tensor(kid).long().cuda()], 4, down="stride", pad="zero",
                                    normalize=False)
    assert torch.equal(out, direct)


def test_vb_peaked_kernels_and_soft_mixture(gpu_device):
    x = landsat_like((2, 5, 96, 96), 23)
    out = kmsr_amd.degrade_batch(dev(x), dev(VB["kgan_b2_96__kernels"]), 4, down="stride", pad="zero", normalize=False)
    assert_close(out.cpu().numpy(), VB["kgan_b2_96__out"], rtol=RTOL, what="kgan")
    x = landsat_like((4, 5, 256, 256), 20)
    out = kmsr_amd.degrade_batch(dev(x), dev(VB["fwd__soft_kernels"]), 4, down="stride", pad="zero", normalize=False)
    assert_close(out.cpu().numpy(), VB["fwd__soft_clean"], rtol=RTOL, what="soft mixture")


def test_gaussian_stream_matches_oracle(gpu_device):
    """The device Philox/Box-Muller stream equals the oracle's definition (integer part exact, transcendental
    part to ~1e-6), for shapes that are not multiples of the wave / row-block sizes."""
    lib = L.load()
    for planes, h, w, seed, off in [(3, 37, 53, 1234, 0), (2, 64, 256, 2 ** 40 + 17, 9), (1, 5, 1, 0, 2 ** 31)]:
        out = torch.empty(planes, h, w).cuda()
        L.check(lib.kmsr_gaussian_eps_f32(L.ptr(out), planes, h, w, seed, off, 0, L.current_stream()))
        want = oracle.gaussian_eps(planes, h, w, seed, off)
        assert_close(out.cpu().numpy(), want, rtol=0, atol=3e-6, what="eps")
    # plane_base: planes [base, base + n) of the stream are exactly that slice of the stream from plane 0
    full = torch.empty(7, 16, 24).cuda()
    part = torch.empty(3, 16, 24).cuda()
    L.check(lib.kmsr_gaussian_eps_f32(L.ptr(full), 7, 16, 24, 99, 5, 0, L.current_stream()))
    L.check(lib.kmsr_gaussian_eps_f32(L.ptr(part), 3, 16, 24, 99, 5, 4, L.current_stream()))
    assert torch.equal(full[4:], part)
    assert_close(part.cpu().numpy(), oracle.gaussian_eps(3, 16, 24, 99, 5, plane_base=4), rtol=0, atol=3e-6, what="eps base")


@pytest.mark.parametrize("flags", [0, L.FLAG_FORCE_GENERIC])
def test_gaussian_noise_is_world_size_invariant(flags, gpu_device):
    """the noise stream is keyed by (seed, offset, GLOBAL sample index), so a batch cut into shards -- two
    contiguous blocks (bench.py / DDP: rank r owns samples r*B..), or the round-robin patch_ids[rank::world] of the per-file
    loops (muti_kernel/train.py:150) -- gives, shard by shard, bit-exactly what one process computes for the whole batch."""
    B, Cc, H, W = 6, 5, 128, 512
    x = dev(landsat_like((B, Cc, H, W), 41))
    kid = torch.tensor([2, 9, 4, 0, 7, 7], dtype=torch.int32)
    kw = dict(down="stride", pad="zero", normalize=False, sigma=dev(K["moe_sigmas"]), noise="gaussian", seed=77, offset=3,
              flags=flags)
    whole = kmsr_amd.degrade_batch(x, dev(K["moe_kernels"]), 4, kernel_idx=kid, **kw)
    # world 2, contiguous blocks
    for r in range(2):
        sl = slice(3 * r, 3 * r + 3)
        part = kmsr_amd.degrade_batch(x[sl], dev(K["moe_kernels"]), 4, kernel_idx=kid[sl], sample_base=3 * r, **kw)
        assert torch.equal(part, whole[sl]), f"contiguous shard {r}"
    # world 3, round robin
    for r in range(3):
        part = kmsr_amd.degrade_batch(x[r::3].contiguous(), dev(K["moe_kernels"]), 4, kernel_idx=kid[r::3], sample_base=r,
                                      sample_step=3, **kw)
        assert torch.equal(part, whole[r::3]), f"round-robin shard {r}"
    # ... and the oracle agrees on a shard addressed globally
    clean = oracle.vb_blur_stride(x[3:].cpu().numpy(), K["moe_kernels"][kid[3:].numpy()], 4)
    want = oracle.add_gaussian_noise(clean, K["moe_sigmas"][kid[3:].numpy()].reshape(-1), seed=77, offset=3, plane_base=3 * Cc)
    assert_close(whole[3:].cpu().numpy(), want, rtol=RTOL, atol=5e-6, what="shard vs oracle")
    # without the base the second shard would repeat the first shard's eps field (round-1 advisor finding)
    p0 = kmsr_amd.degrade_batch(x[3:], dev(K["moe_kernels"]), 4, kernel_idx=kid[3:], **kw)
    assert not torch.equal(p0, whole[3:])
    with pytest.raises(ValueError):
        kmsr_amd.degrade_batch(x[:1], dev(K["moe_kernels"]), 4, kernel_idx=kid[:1], sample_base=2 ** 32, **kw)


@pytest.mark.parametrize("flags", [0, L.FLAG_FORCE_GENERIC])
@pytest.mark.parametrize("rows", [0, 4, 12, 64])
def test_vb_gaussian_noise_fused(flags, rows, gpu_device):
    """out = clean + sigma[row(n),c] * eps(n*C+c, i, j): fused epilogue == oracle composition."""
    B, Cc, H, W = 3, 5, 200, 256
    x = landsat_like((B, Cc, H, W), 40)
    kid = np.array([2, 9, 4], np.int32)
    out = kmsr_amd.degrade_batch(dev(x), dev(K["moe_kernels"]), 4, down="stride", pad="zero", normalize=False,
                                 kernel_idx=dev(kid), sigma=dev(K["moe_sigmas"]), noise="gaussian", seed=77, offset=3,
                                 flags=flags, rows_per_chunk=rows)
    clean = oracle.vb_blur_stride(x, K["moe_kernels"][kid], 4)
    want = oracle.add_gaussian_noise(clean, K["moe_sigmas"][kid].reshape(-1), seed=77, offset=3)
    assert_close(out.cpu().numpy(), want, rtol=RTOL, atol=5e-6, what="gaussian fused")
    resid = out.cpu().numpy() - clean
    assert np.all(np.abs(resid.std(axis=(2, 3)) / K["moe_sigmas"][kid] - 1) < 0.05)


@pytest.mark.parametrize("flags", [0, L.FLAG_FORCE_GENERIC])
def test_pool_noise_fused_and_add_noise(flags, gpu_device):
    """V-C: lr = blurred + noise_pool[idx] (E:65-74) fused into the V-A epilogue."""
    img = landsat_like((4, 5, 256, 256), 41)
    pool = np.random.RandomState(5).randn(50, 5, 32, 32).astype(np.float32)
    idx = np.array([7, 49, 0, 7], np.int64)
    kern = K["kgan_per_band"]
    out = kmsr_amd.degrade_batch(dev(img), dev(kern)[None], 8, noise="pool", noise_pool=dev(pool),
                                 noise_idx=dev(idx), flags=flags)
    blurred = np.stack([oracle.apply_kernel_degradation(img[n], kern, 8) for n in range(4)])
    assert_close(out.cpu().numpy(), oracle.add_pool_noise(blurred, pool, idx), rtol=RTOL, atol=2e-6, what="pool fused")


def test_add_noise_drop_in_stream(gpu_device):
    """add_noise consumes the global NumPy RNG exactly like E_make_train_data.py:72 (seed 42, E:190)."""
    rng = golden("rng_streams.npz")
    pool = np.random.RandomState(6).randn(1000, 5, 32, 32).astype(np.float32)
    blurred = np.random.RandomState(7).randn(8, 5, 32, 32).astype(np.float32)
    np.random.seed(42)
    outs = [kmsr_amd.add_noise(blurred[i], pool) for i in range(8)]
    want_idx = rng["add_noise_idx_seed42_N1000"][:8]
    for i in range(8):
        assert np.array_equal(outs[i], blurred[i] + pool[want_idx[i]])
    np.random.seed(42)
    batch = kmsr_amd.add_noise_batch(blurred, pool)
    assert np.array_equal(batch, np.stack(outs))


def test_noise_pool_construction(gpu_device):
    """D_build_noise_pool.py:56-132 at array level: geo - denoised, seeded crops, stacking."""
    import random
    rs = np.random.RandomState(8)
    pairs = [(rs.randn(5, 64, 80).astype(np.float32), rs.randn(5, 64, 80).astype(np.float32)) for _ in range(3)]
    pairs.insert(1, (np.zeros((5, 16, 16), np.float32), np.zeros((5, 16, 16), np.float32)))   # too small -> skipped (D:102-104)
    pool = kmsr_amd.build_noise_pool_from_arrays(pairs, samples_per_file=2, patch_size=32, seed=42)
    random.seed(42)
    want = []
    for geo, den in pairs:
        if geo.shape[1] < 32:
            continue
        for _ in range(2):
            t, l = random.randint(0, geo.shape[1] - 32), random.randint(0, geo.shape[2] - 32)
            want.append(oracle.noise_crop(geo, den, t, l, 32))
    assert pool.shape == (6, 5, 32, 32)
    assert np.array_equal(pool, np.stack(want))


def test_strided_input_views(gpu_device):
    """hr may be a non-contiguous view (band subset / crop of a larger scene) -- strides travel in the descriptor."""
    big = dev(landsat_like((2, 7, 160, 192), 42))
    view = big[:, 1:6, 16:144, 32:160]          # [2,5,128,128], W-stride 1, 16-B aligned rows
    ker = dev(K["moe_kernels"][5])[None]
    a = kmsr_amd.degrade_batch(view, ker, 4)
    b = kmsr_amd.degrade_batch(view.contiguous(), ker, 4)
    assert torch.equal(a, b)
    view2 = big[:, 1:6, 3:131, 5:133]           # misaligned rows -> generic kernel, same numbers as a contiguous copy
    assert not kmsr_amd.uses_fast_path(view2, ker, 4)
    assert torch.equal(kmsr_amd.degrade_batch(view2, ker, 4), kmsr_amd.degrade_batch(view2.contiguous(), ker, 4,
                                                                                    flags=L.FLAG_FORCE_GENERIC))


def test_rows_per_chunk_invariance(gpu_device):
    """The wave decomposition is a tuning knob.  Odd row chunks accumulate tap rows bottom-up (so neighbouring
    chunks meet at their shared halo rows), hence: for a given chunk height fast == generic bit for bit, and
    different chunk heights agree to fp32 rounding."""
    x = dev(landsat_like((2, 5, 512, 512), 43))
    ker = dev(K["moe_kernels"][:2])
    kw = dict(down="stride", pad="zero", normalize=False)
    ref = kmsr_amd.degrade_batch(x, ker, 4, rows_per_chunk=1000, flags=L.FLAG_FORCE_GENERIC, **kw)   # one chunk: all top-down
    for rows in (4, 8, 20, 32, 128, 1000):
        out = kmsr_amd.degrade_batch(x, ker, 4, rows_per_chunk=rows, flags=L.FLAG_FORCE_FAST, **kw)
        gen = kmsr_amd.degrade_batch(x, ker, 4, rows_per_chunk=rows, flags=L.FLAG_FORCE_GENERIC, **kw)
        assert torch.equal(out, gen), f"rows_per_chunk={rows}: fast != generic"
        assert_close(out.cpu().numpy(), ref.cpu().numpy(), rtol=2e-6, what=f"rows_per_chunk={rows} vs single chunk")
    assert torch.equal(kmsr_amd.degrade_batch(x, ker, 4, rows_per_chunk=1000, flags=L.FLAG_FORCE_FAST, **kw), ref)


# ---------------------------------------------------------------------------------------------------
# Full BASELINE sizes: size-independent properties
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["va_f4", "va_f8", "vb"])
def test_full_size_properties(mode, gpu_device):
    """north-star unit: HR [N,5,1024,1024] -> LR 256^2 (x4) / 128^2 (x8)."""
    N, Cc, H, W = 8, 5, 1024, 1024
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(N, Cc, H, W, device="cuda", generator=g) * 80 + 20
    y = torch.rand(N, Cc, H, W, device="cuda", generator=g) * 80 + 20
    bank = dev(K["moe_kernels"])
    kid = torch.arange(N, dtype=torch.int32, device="cuda") % 10
    if mode == "vb":
        kw = dict(down="stride", pad="zero", normalize=False, kernel_idx=kid)
        f = 4
    else:
        kw = dict(down="avgpool", pad="replicate", normalize=True, kernel_idx=kid)
        f = 4 if mode == "va_f4" else 8
    run = lambda t, **extra: kmsr_amd.degrade_batch(t, bank, f, **kw, **extra)
    ox = run(x, flags=L.FLAG_FORCE_FAST)
    assert tuple(ox.shape) == (N, Cc, H // f, W // f)
    # (1) fast == generic, bit for bit, over the whole batch
    assert torch.equal(ox, run(x, flags=L.FLAG_FORCE_GENERIC))
    # (2) linearity
    lin = run(0.25 * x + 0.5 * y)
    assert_close(lin.cpu().numpy(), (0.25 * ox + 0.5 * run(y)).cpu().numpy(), rtol=RTOL, what="linearity")
    # (3) constants: a sum-1 kernel reproduces a constant image under replicate padding;
    #     under zero padding the interior does and the border is attenuated
    const = torch.full((N, Cc, H, W), 60.0, device="cuda")
    oc = run(const)
    if mode == "vb":
        assert torch.allclose(oc[:, :, 2:-2, 2:-2], torch.full_like(oc[:, :, 2:-2, 2:-2], 60.0), rtol=1e-5)
        assert float(oc[:, :, 0, :].mean()) < 40.0
    else:
        assert torch.allclose(oc, torch.full_like(oc, 60.0), rtol=1e-5)
    # (4) impulse response: a unit impulse at HR (y0,x0) spreads the (flipped) tap table over the LR outputs
    imp = torch.zeros(N, Cc, H, W, device="cuda")
    y0, x0 = 517, 262
    imp[:, :, y0, x0] = 1.0
    oi = run(imp).cpu().numpy()
    kb = K["moe_kernels"][kid.cpu().numpy()]
    if mode != "vb":
        kb = kb / kb.sum(axis=(2, 3), keepdims=True)
    want = np.zeros_like(oi)
    for i in range(oi.shape[2]):
        for j in range(oi.shape[3]):
            if mode == "vb":
                u, v = y0 - f * i + 6, x0 - f * j + 6
                if 0 <= u < 13 and 0 <= v < 13:
                    want[:, :, i, j] = kb[:, :, u, v]
            else:
                us = [y0 - (f * i + dy) + 6 for dy in range(f)]
                vs = [x0 - (f * j + dx) + 6 for dx in range(f)]
                if max(us) >= 0 and min(us) < 13 and max(vs) >= 0 and min(vs) < 13:
                    acc = 0
                    for u in us:
                        for v in vs:
                            if 0 <= u < 13 and 0 <= v < 13:
                                acc = acc + kb[:, :, u, v].astype(np.float64)
                    want[:, :, i, j] = acc / (f * f)
    assert_close(oi, want, rtol=1e-5, atol=1e-9, what="impulse response")
    # (5) mass: with replicate padding and a normalised kernel the global mean is preserved up to border effects
    if mode != "vb":
        assert abs(float(ox.double().mean()) / float(x.double().mean()) - 1) < 1e-3


def test_c_abi_demo_without_torch(gpu_device):
    """examples/c_abi_demo.cpp: a plain C++/HIP program (no torch) drives kmsr_degrade_f32 through include/kmsr.h and
    checks V-A against its own host loop; built by __graft_entry__.build() (make demo)."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kernel-modeling-super-resolution_amd",
                       "c_abi_demo")
    if not os.path.exists(exe):
        pytest.skip("c_abi_demo not built (run __graft_entry__.build())")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "max rel err" in r.stdout and "bad descriptor -> rc -" in r.stdout      # error code + message, no abort
