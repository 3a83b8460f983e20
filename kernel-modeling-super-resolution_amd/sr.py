"""Super-resolution network on the HIP conv kernels (include/kmsr.h: kmsr_conv3x3_*).

The reference repository stops at "make training data" and ships NO SR model, so the
module is build-defined with the shape SURVEY 8a S1 recommends (EDSR-baseline): head conv3x3 C->F; B residual
blocks [conv3x3, ReLU, conv3x3, +skip]; conv3x3 + global skip; log2(scale) x [conv3x3 F->4F, PixelShuffle(2)];
tail conv3x3 F->C.  Its oracle is the same module expressed with stock torch ops (oracle/sr_torch.py) -- parity
for this part is therefore "unpinned by the reference" and stated as HIP-vs-torch on identical weights.

Call surface: an ordinary nn.Module -- `sr = SRNet(...); y = sr(lr)` with lr [B,C,h,w] -> [B,C,4h,4w]; parameters
are standard nn.Conv2d modules so state_dicts interchange with the oracle module.  Forward runs only on the GPU
through the C ABI (no CPU fallback).
"""
import math

import torch
import torch.nn as nn

from . import _lib as L


def pack_conv3x3_weight(weight, transpose_flip=False):
    """[Cout,Cin,3,3] float32 CUDA -> packed buffer consumed by conv3x3 (a permutation + zero padding)."""
    lib = L.load()
    assert weight.is_cuda and weight.dtype == torch.float32 and weight.ndim == 4 and weight.shape[2:] == (3, 3)
    w = weight.detach().contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    cin_p, cout_p = (Cout, Cin) if transpose_flip else (Cin, Cout)
    n = lib.kmsr_conv3x3_packed_weight_floats(cin_p, cout_p)
    packed = torch.empty(n, device=w.device, dtype=torch.float32)
    with torch.cuda.device(w.device):
        L.check(lib.kmsr_conv3x3_pack_weights_f32(L.ptr(w), Cin, Cout, int(transpose_flip), L.ptr(packed),
                                                  L.current_stream()))
    return packed


def pack_conv3x3_weight_wino(weight):
    """[Cout,Cin,3,3] -> Winograd F(2x2,3x3) transformed + packed weights (G g G^T) for conv3x3(..., algo="winograd")."""
    lib = L.load()
    assert weight.is_cuda and weight.dtype == torch.float32 and weight.ndim == 4 and weight.shape[2:] == (3, 3)
    w = weight.detach().contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    packed = torch.empty(lib.kmsr_conv3x3_wino_packed_weight_floats(Cin, Cout), device=w.device, dtype=torch.float32)
    with torch.cuda.device(w.device):
        L.check(lib.kmsr_conv3x3_wino_pack_weights_f32(L.ptr(w), Cin, Cout, L.ptr(packed), L.current_stream()))
    return packed


def pack_conv3x3_weight_wino_vreg(weight):
    """[Cout,Cin,3,3] -> G g G^T packed for conv3x3(..., algo="winograd_vreg") (a lane's four transform columns adjacent)."""
    lib = L.load()
    assert weight.is_cuda and weight.dtype == torch.float32 and weight.ndim == 4 and weight.shape[2:] == (3, 3)
    w = weight.detach().contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    packed = torch.empty(lib.kmsr_conv3x3_winov_packed_weight_floats(Cin, Cout), device=w.device, dtype=torch.float32)
    with torch.cuda.device(w.device):
        L.check(lib.kmsr_conv3x3_winov_pack_weights_f32(L.ptr(w), Cin, Cout, L.ptr(packed), L.current_stream()))
    return packed


def pack_conv3x3_weight_wino4(weight, transpose_flip=False):
    """[Cout,Cin,3,3] -> G g G^T (6x6 points, F(4x4,3x3)) packed for conv3x3(..., algo="winograd4"): per 4-channel step the 36 KB a
    workgroup multiplies, lane-linear per wave and group of 4 transform points (one buffer_load_dwordx4 per lane and group).
    transpose_flip=True packs the layer's DATA-GRADIENT convolution (Cout -> Cin channels, w'[ci,co,u,v] = w[co,ci,2-u,2-v]) in the same
    launch -- no flip / transpose / copy passes (kmsr_conv3x3_wino4_pack_weights_ex_f32)."""
    lib = L.load()
    assert weight.is_cuda and weight.dtype == torch.float32 and weight.ndim == 4 and weight.shape[2:] == (3, 3)
    w = weight.detach().contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    pin, pout = (Cout, Cin) if transpose_flip else (Cin, Cout)
    packed = torch.empty(lib.kmsr_conv3x3_wino4_packed_weight_floats(pin, pout), device=w.device, dtype=torch.float32)
    with torch.cuda.device(w.device):
        L.check(lib.kmsr_conv3x3_wino4_pack_weights_ex_f32(L.ptr(w), Cin, Cout, 1 if transpose_flip else 0, L.ptr(packed), L.current_stream()))
    return packed


def winograd_eligible(cin, cout, stride=1, width=2):
    """Where the Winograd kernel applies/pays: stride 1, even width (patch rows are fetched as aligned float2) and
    enough channels that the 64x16 chunk tiles are not mostly padding."""
    return stride == 1 and width % 2 == 0 and cin >= 16 and cout >= 32


def pick_algo(cin, cout, stride, width, use_winograd=True, plain=False, f4=False):
    """Kernel for a stride-1 3x3 conv of this shape: the persistent LDS-DMA Winograd kernel when rows are 16-byte aligned
    (W % 4 == 0) and the bias vector fits its LDS staging (Cout <= 1024), the register-staged Winograd kernel for other even
    widths, the direct MFMA / VALU kernels otherwise.  plain=True (no residual, no PixelShuffle) lets layers with <= 8 input
    channels take the small-Cin kernel.  f4=True lets aligned layers take the Winograd F(4x4,3x3) kernel (4x fewer MFMA FLOPs than
    direct, 2.25x for F(2x2); same epilogues: residual, PixelShuffle, ReLU mask, channel sums)."""
    if plain and stride == 1 and cin <= 8 and cout >= 32 and width % 4 == 0:
        return "smallci"
    if not (use_winograd and winograd_eligible(cin, cout, stride, width)):
        return "direct"
    if f4 and width % 4 == 0 and cout <= 1024:
        return "winograd4"
    return "winograd_vreg" if width % 4 == 0 and cout <= 1024 else "winograd"


def pack_for(algo, weight):
    if algo == "smallci":                       # the small-Cin kernel reads the [Cout,Cin,3,3] tensor itself
        return weight.detach().to(torch.float32).contiguous()
    return {"direct": pack_conv3x3_weight, "winograd": pack_conv3x3_weight_wino,
            "winograd_vreg": pack_conv3x3_weight_wino_vreg, "winograd4": pack_conv3x3_weight_wino4}[algo](weight)


def conv3x3(x, packed_w, cout, bias=None, relu=False, residual=None, pixel_shuffle=False, stride=1, out=None,
            algo="direct", relu_mask=None, chan_sum=False):
    """out = epilogue(conv3x3(x, w)) with zero padding 1 (semantics: include/kmsr.h).  x [N,Cin,H,W] float32 CUDA.
    algo="direct": packed_w from pack_conv3x3_weight; "winograd" / "winograd_vreg": from pack_conv3x3_weight_wino[_vreg]
    (pick_algo / pack_for choose and pack for a shape).
    Training-step extras of the "winograd_vreg" kernel (kmsr_conv3x3_winov_fwd_ex_f32): relu_mask (out-shaped) zeroes the
    result where relu_mask <= 0 (KMSR_CONV_FLAG_RELU_MASK); chan_sum=True returns (out, sum over (n,y,x) of out per channel)."""
    lib = L.load()
    if not (x.is_cuda and x.dtype == torch.float32 and x.ndim == 4):
        raise ValueError("x must be a 4-D float32 CUDA tensor")
    x = x.contiguous()
    if algo in ("winograd_vreg", "winograd4", "smallci") and x.data_ptr() % 16:
        x = x.clone()                    # a view at an odd storage offset: these kernels fetch 16-byte pieces
    N, Cin, H, W = x.shape
    if algo == "smallci":
        if residual is not None or pixel_shuffle or stride != 1 or relu_mask is not None or chan_sum:
            raise NotImplementedError("the small-Cin kernel has a bias / ReLU epilogue only")
        assert tuple(packed_w.shape) == (cout, Cin, 3, 3) and packed_w.is_contiguous() and packed_w.is_cuda
        d = L.Conv3x3Desc()
        d.N, d.Cin, d.H, d.W, d.Cout, d.stride, d.relu = N, Cin, H, W, int(cout), 1, int(bool(relu))
        if out is None:
            out = torch.empty((N, cout, H, W), device=x.device, dtype=torch.float32)
        else:
            assert tuple(out.shape) == (N, cout, H, W) and out.is_contiguous() and out.dtype == torch.float32
        if bias is not None:
            bias = bias.detach().contiguous()
            assert bias.numel() == cout and bias.is_cuda
        with torch.cuda.device(x.device):
            L.check(lib.kmsr_conv3x3_smallci_fwd_f32(d, L.ptr(x), L.ptr(packed_w), L.ptr(bias), L.ptr(out), L.current_stream()))
        return out
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    d = L.Conv3x3Desc()
    d.N, d.Cin, d.H, d.W, d.Cout, d.stride = N, Cin, H, W, int(cout), int(stride)
    d.relu, d.pixel_shuffle = int(bool(relu)), int(bool(pixel_shuffle))
    if (relu_mask is not None or chan_sum) and algo not in ("winograd_vreg", "winograd4"):
        raise NotImplementedError("relu_mask / chan_sum are epilogues of the winograd_vreg / winograd4 kernels")
    if relu_mask is not None:
        assert residual is None and not relu and not pixel_shuffle
        residual = relu_mask
        d.flags = L.CONV_FLAG_RELU_MASK
    wino = algo in ("winograd", "winograd_vreg", "winograd4")
    if wino and stride != 1:
        raise NotImplementedError("winograd conv needs stride 1")
    need = {"direct": lib.kmsr_conv3x3_packed_weight_floats, "winograd": lib.kmsr_conv3x3_wino_packed_weight_floats,
            "winograd_vreg": lib.kmsr_conv3x3_winov_packed_weight_floats,
            "winograd4": lib.kmsr_conv3x3_wino4_packed_weight_floats}[algo](Cin, int(cout))
    assert packed_w.is_cuda and packed_w.dtype == torch.float32 and packed_w.numel() == need, \
        f"packed weight has {packed_w.numel()} floats, conv {Cin}->{cout} needs {need}"
    oshape = (N, cout // 4, 2 * Ho, 2 * Wo) if pixel_shuffle else (N, cout, Ho, Wo)
    if out is None:
        out = torch.empty(oshape, device=x.device, dtype=torch.float32)
    else:
        assert tuple(out.shape) == oshape and out.is_contiguous() and out.dtype == torch.float32
    if bias is not None:
        bias = bias.detach().contiguous()
        assert bias.numel() == cout and bias.is_cuda
    if residual is not None:
        residual = residual.contiguous()
        assert tuple(residual.shape) == oshape and residual.is_cuda
        if algo == "winograd4" and residual.data_ptr() % 16:
            residual = residual.clone()
    with torch.cuda.device(x.device):
        ex = lib.kmsr_conv3x3_wino4_fwd_ex_f32 if algo == "winograd4" else lib.kmsr_conv3x3_winov_fwd_ex_f32
        if chan_sum:
            nsum = (lib.kmsr_conv3x3_wino4_chan_sum_floats if algo == "winograd4" else lib.kmsr_conv3x3_winov_chan_sum_floats)(d)
            part = torch.empty(nsum, device=x.device, dtype=torch.float32)
            L.check(ex(d, L.ptr(x), L.ptr(packed_w), L.ptr(bias), L.ptr(residual), L.ptr(out), L.ptr(part), L.current_stream()))
            cgs = (cout + 63) // 64                  # part[n][cg][tile][4][64] -> per-channel totals
            return out, part.view(N, cgs, -1, 64).sum(dim=(0, 2)).reshape(-1)[:cout]
        fwd = {"direct": lib.kmsr_conv3x3_fwd_f32, "winograd": lib.kmsr_conv3x3_wino_fwd_f32,
               "winograd4": ex if d.flags else lib.kmsr_conv3x3_wino4_fwd_f32,
               "winograd_vreg": ex if d.flags else lib.kmsr_conv3x3_winov_fwd_f32}[algo]
        if algo in ("winograd_vreg", "winograd4") and d.flags:
            L.check(fwd(d, L.ptr(x), L.ptr(packed_w), L.ptr(bias), L.ptr(residual), L.ptr(out), None, L.current_stream()))
        else:
            L.check(fwd(d, L.ptr(x), L.ptr(packed_w), L.ptr(bias), L.ptr(residual), L.ptr(out), L.current_stream()))
    return out


def conv3x3_wgrad(x, dy, direct=False, wino2=False):
    """dw[co,ci,u,v] = sum_{n,y,x} dy[n,co,y,x] * x[n,ci,y+u-1,x+v-1]  (stride 1, zero padding 1) -> [Cout,Cin,3,3].
    W % 4 == 0, H >= 4 with Cout > 32 run in the Winograd F(4x4,3x3) domain (36 products per 4x4 tile instead of 144); wino2=True
    takes the F(2x2,3x3)-domain kernel instead (even H: 16 products per 2x2 tile), direct=True the tap-by-tap kernel."""
    lib = L.load()
    x = x.contiguous()
    dy = dy.contiguous()
    assert x.is_cuda and dy.is_cuda and x.dtype == dy.dtype == torch.float32
    N, Cin, H, W = x.shape
    assert dy.shape[0] == N and tuple(dy.shape[2:]) == (H, W)
    Cout = dy.shape[1]
    d = L.Conv3x3Desc()
    d.N, d.Cin, d.H, d.W, d.Cout, d.stride = N, Cin, H, W, Cout, 1
    d.flags = L.CONV_FLAG_WGRAD_DIRECT if direct else (L.CONV_FLAG_WGRAD_WINO2 if wino2 else 0)
    nbytes = lib.kmsr_conv3x3_wgrad_workspace_bytes(d)
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
    dw = torch.empty((Cout, Cin, 3, 3), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        L.check(lib.kmsr_conv3x3_wgrad_f32(d, L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(ws), nbytes, L.current_stream()))
    return dw


class _TrainPacks:
    """Packed weights of the training path, per parameter VERSION: (algo, direction) -> packed buffer, rebuilt when the
    parameter is modified in place (optimizer step) or the entry's tensor has been freed.  Forward and data-gradient
    launches of one step, gradient accumulation micro-steps and no-grad evaluation passes share one pack."""
    _store = {}

    @classmethod
    def get(cls, w, algo, flip):
        import weakref
        key = (w.data_ptr(), tuple(w.shape), algo, flip)
        ent = cls._store.get(key)
        if ent is not None and ent[0] == w._version and ent[1]() is w:
            return ent[2]
        src = w.detach()
        if flip and algo == "winograd4":               # the pack launch reads the forward tensor through the flipped / transposed index map
            packed = pack_conv3x3_weight_wino4(src, transpose_flip=True)
        else:
            if flip:                                   # data gradient = the same convolution with w'[ci,co,u,v] = w[co,ci,2-u,2-v]
                src = src.flip(2, 3).transpose(0, 1).contiguous()
            packed = pack_for(algo, src)
        if len(cls._store) > 4096:                     # parameters that went away (their storage may be reused under another key)
            cls._store = {k: e for k, e in cls._store.items() if e[1]() is not None}
        cls._store[key] = (w._version, weakref.ref(w), packed)
        return packed


def pixel_unshuffle2_sum(g):
    """g [N,C,2H,2W] -> (gu [N,4C,H,W], per-channel sums [4C]) in one pass (kmsr_pixel_unshuffle2_sum_f32)."""
    lib = L.load()
    g = g.contiguous()
    N, C, H2, W2 = g.shape
    H, W = H2 // 2, W2 // 2
    if W % 4 or g.data_ptr() % 16 or not g.is_cuda or g.dtype != torch.float32:
        gu = torch.nn.functional.pixel_unshuffle(g, 2).contiguous()
        return gu, gu.sum(dim=(0, 2, 3))
    gu = torch.empty((N, 4 * C, H, W), device=g.device, dtype=torch.float32)
    part = torch.empty(lib.kmsr_pixel_unshuffle2_sum_partials(N, C, H), device=g.device, dtype=torch.float32)
    with torch.cuda.device(g.device):
        L.check(lib.kmsr_pixel_unshuffle2_sum_f32(L.ptr(g), N, C, H, W, L.ptr(gu), L.ptr(part), L.current_stream()))
    return gu, part.view(-1, 4 * C).sum(dim=0)


def conv3x3_smallci_unshuffle(x, w, bias=None, sums="kernel"):
    """conv3x3(x [N,Cin<=8,H,W], w [Cout<=64,Cin,3,3]) stored pixel-unshuffled -> (y [N,4 Cout,H/2,W/2], per-channel sums [4 Cout])
    (kmsr_conv3x3_smallci_unshuffle_fwd_f32): y[n, 4 co + 2 (i & 1) + (j & 1), i >> 1, j >> 1] = conv(x, w)[n, co, i, j].
    sums="kernel": accumulated by the launch itself (fixed order); "closed_form": the launch only stores (2.5 instead of 4.7 ms at config 3)
    and the sums come from unshuffled_conv_channel_sums (bias must be None); None: no sums."""
    lib = L.load()
    x = x.contiguous()
    w = w.detach().to(torch.float32).contiguous()
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    d = L.Conv3x3Desc()
    d.N, d.Cin, d.H, d.W, d.Cout, d.stride = N, Cin, H, W, Cout, 1
    y = torch.empty((N, 4 * Cout, H // 2, W // 2), device=x.device, dtype=torch.float32)
    if sums != "kernel":
        assert sums in (None, "closed_form") and (sums is None or bias is None)
        with torch.cuda.device(x.device):
            L.check(lib.kmsr_conv3x3_smallci_unshuffle_fwd_f32(d, L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), None, None, 0, L.current_stream()))
        return y, (unshuffled_conv_channel_sums(x, w) if sums else None)
    csum = torch.empty(4 * Cout, device=x.device, dtype=torch.float32)
    nbytes = lib.kmsr_conv3x3_smallci_unshuffle_workspace_bytes(d)
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        L.check(lib.kmsr_conv3x3_smallci_unshuffle_fwd_f32(d, L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), L.ptr(csum), L.ptr(ws), nbytes,
                                                           L.current_stream()))
    return y, csum


_PARITY_MAPS = {}


def _parity_maps(H, W, device):
    """A [3,2,H], B [3,2,W]: A[a, p, y'] = 1 when input row y' is reached through tap row a by an output row of parity p that lies inside the
    image (output row y = y' - (a - 1)); the same for columns."""
    key = (H, W, str(device))
    if key not in _PARITY_MAPS:
        def one(n):
            m = torch.zeros(3, 2, n, dtype=torch.float32)
            for a in range(3):
                for yi in range(n):
                    y = yi - (a - 1)
                    if 0 <= y < n:
                        m[a, y & 1, yi] = 1.0
            return m
        _PARITY_MAPS[key] = (one(H).to(device), one(W).to(device))
    return _PARITY_MAPS[key]


def unshuffled_conv_channel_sums(x, w):
    """sums[4 co + 2 p + q] = sum over n and the output pixels (y, x) of parity (p, q) of conv3x3(x, w)[n, co, y, x] (zero padding 1, no bias),
    WITHOUT the convolution: sum_p conv(x, w)[co, p] = sum_{ci,a,b} w[co,ci,a,b] * S[ci,a,b,p,q], where S is the sum of x[ci] over the input
    pixels that tap (a, b) reaches from an in-image output pixel of that parity -- parity-class sums of x minus a border row / column.  One
    reduction over the batch + two thin matrix products on the [Cin,H,W] image: 0.3 ms at config 3 where the sums inside the un-shuffling
    conv cost 2.2 ms of spills.  fp32, fixed order (deterministic); differs from the in-kernel sums by re-association only."""
    N, C, H, W = x.shape
    A, B = _parity_maps(H, W, x.device)
    G = x.sum(dim=0)                                                   # [C,H,W]
    T = torch.matmul(G.reshape(C * H, W), B.reshape(6, W).t())         # [C*H, (b,q)]
    S = torch.matmul(A.reshape(6, H), T.reshape(C, H, 6))              # [C, (a,p), (b,q)]
    S = S.reshape(C, 3, 2, 3, 2)
    return torch.einsum("ocab,capbq->opq", w.to(torch.float32), S).reshape(-1)


def smallci_unshuffle_eligible(cin, cout, H, W):
    return cin <= 8 and 32 <= cout <= 64 and H % 2 == 0 and W % 4 == 0 and H * W * 4 * 64 < 2 ** 31


def _grad_sum(g):
    """Per-channel sum of a gradient tensor: taken from the producing launch's epilogue (tag set by _tag_sum on exactly this
    tensor object and storage version) when there is one, else one reduction pass."""
    tag = getattr(g, "_kmsr_chan_sum", None)
    if tag is not None and tag[1] == g.data_ptr() and tag[2] == g._version and tag[0].numel() == g.shape[1]:
        return tag[0]
    return g.sum(dim=(0, 2, 3))


def _tag_sum(t, sums):
    if sums is not None:
        t._kmsr_chan_sum = (sums, t.data_ptr(), t._version)
    return t


class L1LossFunction(torch.autograd.Function):
    """mean |a - b| with the gradient w.r.t. a produced by the forward's own pass (kmsr_l1_loss_f32: value + sign(a - b) / n in one launch over
    the two tensors; torch's l1_loss and its backward walk them five times).  b gets no gradient (the target).  backward rescales the stored
    gradient in place by the upstream gradient, read on the device (kmsr_scale_by_device_scalar_f32: a no-op launch when it is 1)."""

    @staticmethod
    def forward(ctx, a, b):
        lib = L.load()
        L.require_gpu()
        assert a.shape == b.shape and a.dtype == torch.float32 and b.dtype == torch.float32 and a.is_cuda and b.is_cuda
        a, b = a.contiguous(), b.contiguous()
        loss = torch.empty((), device=a.device, dtype=torch.float32)
        grad = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        ws = torch.empty(lib.kmsr_l1_loss_workspace_floats(), device=a.device, dtype=torch.float32)
        with torch.cuda.device(a.device):
            L.check(lib.kmsr_l1_loss_f32(L.ptr(a), L.ptr(b), a.numel(), L.ptr(loss), L.ptr(grad), L.ptr(ws), L.current_stream()))
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, g):
        grad, ctx.grad = ctx.grad, None
        if grad is None:
            return None, None
        lib = L.load()
        g = g.detach().to(torch.float32).contiguous()
        with torch.cuda.device(grad.device):
            L.check(lib.kmsr_scale_by_device_scalar_f32(L.ptr(grad), grad.numel(), L.ptr(g), L.current_stream()))
        return grad, None


def l1_loss(a, b):
    """torch.nn.functional.l1_loss(a, b) (mean reduction, gradient to `a` only) on the fused HIP kernel."""
    return L1LossFunction.apply(a, b)


class Conv3x3Function(torch.autograd.Function):
    """y = [relu](conv3x3(x, w) + b) with the HIP kernels in both directions (stride 1):
    forward  kmsr_conv3x3[_wino|_winov]_fwd_f32, backward  dx = conv3x3(g, flip-transposed w) (same kernels), dw =
    kmsr_conv3x3_wgrad_f32, db = per-channel sum of g (from the epilogue of the launch that produced g when that was one of
    ours).  ReLU's mask is applied to the incoming gradient from the saved output.
    Forward and data gradient take the Winograd kernel where it is eligible (F->F and F->4F layers)."""
    use_winograd = True
    use_winograd4 = True

    @staticmethod
    def _run(x, w, bias, relu, residual=None, pixel_shuffle=False, flip=False, relu_mask=None, chan_sum=False):
        """w: the layer's [Cout,Cin,3,3] parameter; flip=True runs the data-gradient convolution (Cin <-> Cout)."""
        cout, cin = (w.shape[1], w.shape[0]) if flip else (w.shape[0], w.shape[1])
        # forward and data-gradient launches of the F -> F and F -> 4F layers: Winograd F(4x4,3x3) (it has every epilogue of the F(2x2)
        # kernel: residual, PixelShuffle, ReLU mask, channel sums)
        algo = pick_algo(cin, cout, 1, x.shape[-1], Conv3x3Function.use_winograd,
                         plain=residual is None and not pixel_shuffle and relu_mask is None, f4=Conv3x3Function.use_winograd4)
        packed = _TrainPacks.get(w, algo, flip)
        if algo not in ("winograd_vreg", "winograd4") or pixel_shuffle:     # the training extras exist in these two kernels only (never with PS)
            # no fused sums here: (y, None) -- whoever needs the sums of y reduces it then (often nobody does)
            y = conv3x3(x, packed, cout, bias=bias, relu=relu, residual=residual, pixel_shuffle=pixel_shuffle, algo=algo)
            if relu_mask is not None:
                y = torch.ops.aten.threshold_backward(y, relu_mask, 0.0)
            return (y, None) if chan_sum else y
        return conv3x3(x, packed, cout, bias=bias, relu=relu, residual=residual, pixel_shuffle=pixel_shuffle, algo=algo,
                       relu_mask=relu_mask, chan_sum=chan_sum)

    @staticmethod
    def forward(ctx, x, weight, bias, relu, residual=None, pixel_shuffle=False):
        """residual (same shape as the conv output) is added and PixelShuffle(2) applied in the kernel's epilogue, so the
        training graph has no separate add / permutation passes in the forward direction."""
        if residual is not None and (relu or pixel_shuffle):
            # the epilogue is relu(conv + b) + residual (kmsr.h): the ReLU mask could not be taken from the saved output, and
            # the ABI rejects residual + pixel_shuffle; no layer of the SR / selector nets needs either combination
            raise ValueError("Conv3x3Function: residual cannot be combined with relu or pixel_shuffle")
        x = x.contiguous()
        y = Conv3x3Function._run(x, weight, bias, relu, residual, pixel_shuffle)
        ctx.relu = relu
        ctx.pixel_shuffle = pixel_shuffle
        ctx.has_residual = residual is not None
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        db = None
        if ctx.pixel_shuffle:
            g, s = pixel_unshuffle2_sum(g)                    # gradient in the conv-output layout + its channel sums, one pass
            _tag_sum(g, s)
        g = g.contiguous()
        if ctx.relu:
            g = torch.ops.aten.threshold_backward(g, y, 0.0)      # g where y > 0 else 0, one pass (no mask tensor)
        dx = dw = dres = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _grad_sum(g)
        if ctx.needs_input_grad[0]:
            # the data gradient's own per-channel sums ride along: they are the bias gradient of the layer below
            dx, s = Conv3x3Function._run(g, weight, None, False, flip=True, chan_sum=True)
            _tag_sum(dx, s)
        if ctx.needs_input_grad[1]:
            dw = conv3x3_wgrad(x, g)
        if ctx.has_residual and ctx.needs_input_grad[4]:
            dres = g                                          # y = conv(x) + residual
        return dx, dw, db, None, dres, None


# ------------------------------------------------------------------------------------------------------------------------
# Stride-2 conv3x3 (pad 1) with a backward at the MINIMUM work (round 4; the selector / condition-encoder nets of the reference's trainers,
# train_gemini.py:18-31, muti_kernel/networks.py:21-33).  Until round 3 the training path ran the stride-1 conv and kept every second
# pixel: 4x the work in forward, data gradient and weight gradient.  Both gradients are stride-1 problems in pixel-(un)shuffled
# coordinates, so the existing stride-1 kernels serve them exactly:
#   y[i, j]       = sum_{u,v} w[u, v] x[2i + u - 1, 2j + v - 1]
#   dx[2I + a, .] = sum over (a, s) -> u in {(0,1)->1, (1,1)->2, (1,2)->0} of w[u, .] g[I + s - 1, .]
#                 = PixelShuffle2( conv3x3_stride1(g, W') ),  W'[4 ci + 2a + b, co, s, t] = w[co, ci, u(a,s), v(b,t)]  (other taps zero)
#                   -> ONE launch of the F(4x4,3x3) Winograd kernel with its fused PixelShuffle store: 9 products per LR pixel and
#                      (co, ci) pair, the minimum;
#   dw[co,ci,u,v] = sum g[co,i,j] x[ci, 2i + u - 1, 2j + v - 1] = dw1[co, 4 ci + 2a + b, s, t] with u -> (a, s) in {0->(1,0), 1->(0,1), 2->(1,1)},
#                   dw1 = the stride-1 weight gradient between xu = PixelUnshuffle2(x) and g (F(4x4,3x3) domain where eligible).
def s2_dgrad_weights(w):
    """w [Cout,Cin,3,3] of a stride-2 conv -> W' [4 Cin, Cout, 3, 3]: dx = pixel_shuffle(conv2d(g, W', padding=1), 2)[..., :H, :W]."""
    Cout, Cin = w.shape[0], w.shape[1]
    wp = w.new_zeros((Cin, 2, 2, Cout, 3, 3))
    umap = {(0, 1): 1, (1, 1): 2, (1, 2): 0}
    for (a, s_), u in umap.items():
        for (b, t_), v in umap.items():
            wp[:, a, b, :, s_, t_] = w[:, :, u, v].transpose(0, 1)
    return wp.reshape(4 * Cin, Cout, 3, 3)


def s2_wgrad_gather(dw1, cin):
    """dw1 [Cout, 4 Cin, 3, 3] = stride-1 weight gradient between pixel_unshuffle(x, 2) and g -> dw [Cout, Cin, 3, 3] of the stride-2 conv."""
    cout = dw1.shape[0]
    d = dw1.reshape(cout, cin, 2, 2, 3, 3)
    amap = {0: (1, 0), 1: (0, 1), 2: (1, 1)}                 # u -> (phase a, tap s)
    dw = dw1.new_empty((cout, cin, 3, 3))
    for u, (a, s_) in amap.items():
        for v, (b, t_) in amap.items():
            dw[:, :, u, v] = d[:, :, a, b, s_, t_]
    return dw


class Conv3x3S2Function(torch.autograd.Function):
    """y = conv3x3(x, w, stride 2, padding 1) + b on the HIP kernels in both directions: forward kmsr_conv3x3_fwd_f32 (stride 2), data
    gradient = one F(4x4,3x3) launch with the fused PixelShuffle store on re-indexed weights, weight gradient = the stride-1 kernel on
    the pixel-unshuffled input (see the block comment above).  ReLU / BatchNorm stay with the caller."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        y = conv3x3(x, _TrainPacks.get(weight, "direct", False), weight.shape[0], bias=bias, stride=2, algo="direct")
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        N, Cin, H, W = x.shape
        Cout, Ho, Wo = w.shape[0], g.shape[2], g.shape[3]
        dx = dw = db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = g.sum(dim=(0, 2, 3))
        if ctx.needs_input_grad[0]:
            wp = s2_dgrad_weights(w.detach())
            algo = pick_algo(Cout, 4 * Cin, 1, Wo, Conv3x3Function.use_winograd, f4=Conv3x3Function.use_winograd4)
            dxf = conv3x3(g, pack_for(algo, wp), 4 * Cin, pixel_shuffle=True, algo=algo)          # [N, Cin, 2 Ho, 2 Wo]
            dx = dxf if (2 * Ho, 2 * Wo) == (H, W) else dxf[:, :, :H, :W].contiguous()
        if ctx.needs_input_grad[1]:
            xe = x if (2 * Ho, 2 * Wo) == (H, W) else torch.nn.functional.pad(x, (0, 2 * Wo - W, 0, 2 * Ho - H))
            xu = torch.nn.functional.pixel_unshuffle(xe, 2).contiguous()                            # [N, 4 Cin, Ho, Wo]
            dw = s2_wgrad_gather(conv3x3_wgrad(xu, g), Cin)
        return dx, dw, db


class UpTailFunction(torch.autograd.Function):
    """out = tail(PixelShuffle2(up(h))) -- the last upsampler conv (F -> 4F, PixelShuffle fused into its store) and the tail conv
    (F -> bands) as ONE autograd node, so that the tail's data gradient can be written directly in the un-shuffled layout the up conv
    takes as its output gradient (kmsr_conv3x3_smallci_unshuffle_fwd_f32): the separate pixel_unshuffle2_sum pass over the HR-sized
    gradient map (2 x 8.6 GB at config 3) disappears.  The up conv's bias gradient -- the per-channel sums of that map -- comes in closed form
    from the loss gradient's parity / border sums (unshuffled_conv_channel_sums).  Same arithmetic as the two Conv3x3Function nodes (the
    un-shuffle is a permutation; the bias-gradient sums are re-associated)."""

    @staticmethod
    def forward(ctx, h, w_up, b_up, w_tail, b_tail):
        h = h.contiguous()
        y2 = Conv3x3Function._run(h, w_up, b_up, False, None, True)
        out = Conv3x3Function._run(y2, w_tail, b_tail, False)
        ctx.save_for_backward(h, w_up, y2, w_tail)
        ctx.has_bias = (b_up is not None, b_tail is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        h, w_up, y2, w_tail = ctx.saved_tensors
        g = g.contiguous()
        db_tail = _grad_sum(g) if ctx.has_bias[1] and ctx.needs_input_grad[4] else None
        dw_tail = conv3x3_wgrad(y2, g) if ctx.needs_input_grad[3] else None
        # tail data gradient = conv3x3(g, w') with w'[ci,co,u,v] = w_tail[co,ci,2-u,2-v], stored un-shuffled + its channel sums
        wt = w_tail.detach().flip(2, 3).transpose(0, 1).contiguous()
        want_db_up = ctx.has_bias[0] and ctx.needs_input_grad[2]
        gu, s = conv3x3_smallci_unshuffle(g, wt, sums="closed_form" if want_db_up else None)
        db_up = s if want_db_up else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx, sx = Conv3x3Function._run(gu, w_up, None, False, flip=True, chan_sum=True)
            _tag_sum(dx, sx)
        dw_up = conv3x3_wgrad(h, gu) if ctx.needs_input_grad[1] else None
        return dx, dw_up, db_up, dw_tail, db_tail


class ResBlockFunction(torch.autograd.Function):
    """y = h + conv2(relu(conv1(h))) as ONE autograd node (HIP kernels in both directions).  Backward, per block: two data-
    gradient launches whose epilogues apply the ReLU mask (dt = dgrad2(g) where t > 0), add the skip path's gradient
    (dh = dgrad1(dt) + g) and emit the per-channel sums that are the two bias gradients; two weight-gradient launches.
    No elementwise or reduction pass over a feature map outside those four launches."""

    @staticmethod
    def forward(ctx, h, w1, b1, w2, b2):
        h = h.contiguous()
        t = Conv3x3Function._run(h, w1, b1, True)
        y = Conv3x3Function._run(t, w2, b2, False, residual=h)
        ctx.save_for_backward(h, t, w1, w2)
        ctx.has_bias = (b1 is not None, b2 is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        h, t, w1, w2 = ctx.saved_tensors
        db2 = _grad_sum(g) if ctx.has_bias[1] and ctx.needs_input_grad[4] else None
        g = g.contiguous()
        dw2 = conv3x3_wgrad(t, g) if ctx.needs_input_grad[3] else None
        dt, s_dt = Conv3x3Function._run(g, w2, None, False, flip=True, relu_mask=t, chan_sum=True)
        dw1 = conv3x3_wgrad(h, dt) if ctx.needs_input_grad[1] else None
        db1 = (s_dt if s_dt is not None else dt.sum(dim=(0, 2, 3))) if ctx.has_bias[0] and ctx.needs_input_grad[2] else None
        dh = None
        if ctx.needs_input_grad[0]:
            dh, s_dh = Conv3x3Function._run(dt, w1, None, False, flip=True, residual=g, chan_sum=True)
            _tag_sum(dh, s_dh)
        return dh, dw1, db1, dw2, db2


class _PackCache:
    """Packed weights of one nn.Conv2d, rebuilt when the parameter is modified in place or replaced."""

    def __init__(self):
        self.key = None
        self.packed = None

    def get(self, conv, algo="direct"):
        w = conv.weight
        key = (w.data_ptr(), w._version, w.device, algo)
        if key != self.key:
            self.packed = pack_for(algo, w)
            self.key = key
        return self.packed


def compose_upsampler(up1, up2, tail):
    """Collapse  tail(PS2(up2(PS2(up1(h)))))  -- linear, no activation in between -- into one 5x5 convolution on the LR
    grid followed by PixelShuffle(4):  -> (w5 [16*bands, F, 5, 5], b5 [16*bands]) float32 on the CPU.

    The composite taps are read off the chain's response to unit impulses (and to a zero image for the bias), evaluated
    in float64 so the only rounding is the final cast; host-side weight preparation like pack_conv3x3_weight, done once
    per weight version.  Exact wherever tail and up2 read only in-image data: HR pixels >= 3 from the border (up1's zero padding of the
    LR input is reproduced by the 5x5 kernel's own zero padding)."""
    import torch.nn.functional as Fn
    F, bands = up1.in_channels, tail.out_channels
    cvt = lambda t: t.detach().to("cpu", torch.float64)

    def chain(h, with_bias):
        for cv in (up1, up2):
            h = Fn.pixel_shuffle(Fn.conv2d(h, cvt(cv.weight), cvt(cv.bias) if with_bias and cv.bias is not None else None,
                                           padding=1), 2)
        return Fn.conv2d(h, cvt(tail.weight), cvt(tail.bias) if with_bias and tail.bias is not None else None, padding=1)

    S, c0 = 11, 5
    imp = torch.zeros(F, F, S, S, dtype=torch.float64)
    imp[torch.arange(F), torch.arange(F), c0, c0] = 1.0
    r = chain(imp, False)                                           # [ci, band, 4S, 4S]
    # out[band, 4Y+a, 4X+b] = w5[band*16+a*4+b, ci, dy, dx] with Y = c0 + 2 - dy, X = c0 + 2 - dx
    r = r[:, :, 4 * (c0 - 2):4 * (c0 + 3), 4 * (c0 - 2):4 * (c0 + 3)].reshape(F, bands, 5, 4, 5, 4)   # [ci,band,Yi,a,Xi,b]
    r = r.flip(2).flip(4)                                           # Yi = 4 - dy -> dy
    w5 = r.permute(1, 3, 5, 0, 2, 4).reshape(bands * 16, F, 5, 5)
    z = chain(torch.zeros(1, F, S, S, dtype=torch.float64), True)
    b5 = z[0, :, 4 * c0:4 * c0 + 4, 4 * c0:4 * c0 + 4].reshape(bands * 16)
    return w5.float().contiguous(), b5.float().contiguous()


def conv5x5_ps4(x, packed_w, cout, bias=None, out=None):
    """out[n, co>>4, 4y+((co>>2)&3), 4x+(co&3)] = bias[co] + conv5x5(x, w)[co] (zero pad 2): kmsr_conv5x5_ps4_fwd_f32."""
    lib = L.load()
    if not (x.is_cuda and x.dtype == torch.float32 and x.ndim == 4):
        raise ValueError("x must be a 4-D float32 CUDA tensor")
    x = x.contiguous()
    N, Cin, H, W = x.shape
    assert packed_w.is_cuda and packed_w.numel() == lib.kmsr_conv5x5_ps4_packed_weight_floats(Cin, int(cout))
    oshape = (N, cout // 16, 4 * H, 4 * W)
    if out is None:
        out = torch.empty(oshape, device=x.device, dtype=torch.float32)
    else:
        assert tuple(out.shape) == oshape and out.is_contiguous() and out.dtype == torch.float32
    if bias is not None:
        assert bias.is_cuda and bias.numel() == cout and bias.is_contiguous()
    with torch.cuda.device(x.device):
        L.check(lib.kmsr_conv5x5_ps4_fwd_f32(N, Cin, H, W, int(cout), L.ptr(x), L.ptr(packed_w), L.ptr(bias), L.ptr(out),
                                             L.current_stream()))
    return out


class _FusedUpsampler:
    """Inference-time executor of the collapsed x4 upsampler (SRNet.fuse_upsampler).

    Interior: one conv5x5_ps4 launch on the composed weights (compose_upsampler).  The outer HR frame (one LR row / column = 4 HR
    pixels; the zero padding of up2 and tail reaches 3 of them) is a DIFFERENT but still linear 5x5 map of the LR features, one
    per border class (top / bottom / left / right / four corners): the intermediate layers' zero padding only truncates paths.
    Those 8 weight sets are read off the layer-by-layer HIP kernels' responses to unit impulses on a 5x5 LR canvas (every LR
    pixel of that canvas sits in exactly one class), once per weight version; at run time ONE launch of kmsr_upsampler_frame_f32
    evaluates them for the 2W + 2(H-2) border LR pixels of every image and writes their 4x4 HR blocks over the interior launch's
    output.  Images smaller than MIN_SIZE LR pixels on a side run layer by layer (SRNet._forward_infer)."""
    MAX_BANDS = 8    # kmsr_upsampler_frame_f32 keeps a band's 16 sub-pixel sums in registers for <= 8 bands (FMAXB); more bands run layer by layer
    MIN_SIZE = 8     # every border class must see an untruncated neighbourhood on its far side (>= 5 would do; 8 keeps tiles sane)

    def __init__(self):
        self.key = None

    @staticmethod
    def _pack5(w5):
        lib = L.load()
        packed = torch.empty(lib.kmsr_conv5x5_ps4_packed_weight_floats(w5.shape[1], w5.shape[0]), device=w5.device, dtype=torch.float32)
        with torch.cuda.device(w5.device):
            L.check(lib.kmsr_conv5x5_ps4_pack_weights_f32(L.ptr(w5.contiguous()), w5.shape[1], w5.shape[0], L.ptr(packed), L.current_stream()))
        return packed

    def _prepare(self, net):
        up1, up2 = net.up
        tensors = [t for cv in (up1, up2, net.tail) for t in (cv.weight, cv.bias) if t is not None]
        key = tuple((t.data_ptr(), t._version, t.device) for t in tensors) + (net.use_winograd,)
        if key == self.key:
            return
        dev = up1.weight.device
        w5, b5 = compose_upsampler(up1, up2, net.tail)
        w5 = w5.to(dev)
        self.cout5 = w5.shape[0]
        self.w5 = self._pack5(w5)
        self.b5 = b5.to(dev)
        # layer-by-layer weights for the impulse probing of the border classes
        self.layers = [(cv.weight.detach().contiguous(), cv.bias.detach().contiguous() if cv.bias is not None else None, {})
                       for cv in (up1, up2, net.tail)]              # (w, b, packed weights per algorithm)
        self._border_sets(net, dev)
        self.key = key

    def _border_sets(self, net, dev):
        """5x5 weights + bias of the 8 border classes from ONE run of the three layers on the 25 F unit impulses of a 5x5 canvas."""
        F, bands = net.up[0].in_channels, net.tail.out_channels
        S = 5
        imp = torch.zeros(S * S * F + 1, F, S, S, device=dev)          # image (p*S + q)*F + ci has a 1 at (ci, p, q); the last one is all zero
        idx = torch.arange(S * S * F, device=dev)
        imp[idx, idx % F, (idx // F) // S, (idx // F) % S] = 1.0
        r = self._chain(imp, net.use_winograd)                          # [25F+1, bands, 20, 20]
        zero = r[-1]
        r = (r[:-1] - zero).view(S, S, F, bands, 4 * S, 4 * S)          # [p, q, ci, band, Y, X]
        order = {(-1, 0): 0, (1, 0): 1, (0, -1): 2, (0, 1): 3, (-1, -1): 4, (-1, 1): 5, (1, -1): 6, (1, 1): 7}    # kmsr.h class ids
        self.wcls = torch.zeros(8, F, 25, bands * 16, device=dev)
        self.bcls = torch.zeros(8, bands * 16, device=dev)
        for (ri, ci_), c in order.items():
            i0, j0 = 2 + 2 * ri, 2 + 2 * ci_                            # the canvas pixel of this class
            w = torch.zeros(bands, 4, 4, F, 5, 5, device=dev)           # [band, a, b, ci, dy, dx]
            for dy in range(5):
                for dx in range(5):
                    p, q = i0 + dy - 2, j0 + dx - 2
                    if 0 <= p < S and 0 <= q < S:
                        w[:, :, :, :, dy, dx] = r[p, q, :, :, 4 * i0:4 * i0 + 4, 4 * j0:4 * j0 + 4].permute(1, 2, 3, 0)
            self.wcls[c] = w.permute(3, 4, 5, 0, 1, 2).reshape(F, 25, bands * 16)
            self.bcls[c] = zero[:, 4 * i0:4 * i0 + 4, 4 * j0:4 * j0 + 4].reshape(bands * 16)
        self.bands = bands

    def _chain(self, s, use_wino):
        """up1 -> PS -> up2 -> PS -> tail on a batch [M,F,R,L] with the HIP 3x3 kernels."""
        for i, (w, b, packs) in enumerate(self.layers):
            algo = pick_algo(w.shape[1], w.shape[0], 1, s.shape[-1], use_wino)
            if algo not in packs:
                packs[algo] = pack_for(algo, w)
            s = conv3x3(s, packs[algo], w.shape[0], bias=b, pixel_shuffle=i < 2, algo=algo)
        return s

    def __call__(self, net, h):
        self._prepare(net)
        N, F, H, W = h.shape
        assert min(H, W) >= self.MIN_SIZE
        y = conv5x5_ps4(h, self.w5, self.cout5, bias=self.b5)
        lib = L.load()
        with torch.cuda.device(h.device):
            L.check(lib.kmsr_upsampler_frame_f32(L.ptr(h), N, F, H, W, self.bands, L.ptr(self.wcls), L.ptr(self.bcls), L.ptr(y),
                                                 L.current_stream()))
        return y


class ResBlock(nn.Module):
    def __init__(self, n_feats):
        super().__init__()
        self.conv1 = nn.Conv2d(n_feats, n_feats, 3, padding=1)
        self.conv2 = nn.Conv2d(n_feats, n_feats, 3, padding=1)


class SRNet(nn.Module):
    """EDSR-baseline-shaped SR network (x2^k).  Default: 64 features, 16 residual blocks, x4 -> 1.52 M parameters,
    3.97 MFLOP per LR pixel at 3 bands."""

    def __init__(self, n_colors=3, n_feats=64, n_resblocks=16, scale=4):
        super().__init__()
        assert scale in (2, 4, 8), "scale must be a power of two"
        self.n_colors, self.n_feats, self.scale = n_colors, n_feats, scale
        self.head = nn.Conv2d(n_colors, n_feats, 3, padding=1)
        self.body = nn.ModuleList([ResBlock(n_feats) for _ in range(n_resblocks)])
        self.body_tail = nn.Conv2d(n_feats, n_feats, 3, padding=1)
        self.up = nn.ModuleList([nn.Conv2d(n_feats, 4 * n_feats, 3, padding=1) for _ in range(int(math.log2(scale)))])
        self.tail = nn.Conv2d(n_feats, n_colors, 3, padding=1)
        self._caches = {}
        self.use_winograd = True    # inference path: Winograd MFMA kernels for the F->F / F->4F convs
        self.use_winograd4 = True   # inference path: F(4x4,3x3) for the aligned F->F / F->4F layers; False = F(2x2,3x3)
        # training path: last upsampler conv + tail as one autograd node (UpTailFunction: the tail's data gradient is stored un-shuffled by the
        # small-Cin kernel itself -- lanes own the pixels of ONE sub-pixel plane, so the store is a full line -- and the up conv's bias gradient comes
        # in closed form from the loss gradient's parity / border sums): no pixel_unshuffle2_sum pass over the 8.6 GB map.  85.5-85.9 vs 88.9 ms per
        # config-3 step (A/B in one process, tools/time_train_ab.py; the first form of round 4 -- half-line stores, sums in registers -- was 2.5 ms slower).
        self.fuse_up_tail_backward = True
        self.fuse_upsampler = True   # inference path, x4 only: up1/PS/up2/PS/tail (linear) run as one 5x5 conv + PS4 + the border-class frame launch
        #                              (same outputs, tests/test_fullsize_gpu.py; False = three conv3x3 launches)
        self._fused_up = _FusedUpsampler()

    def _packed(self, name, conv, algo="direct"):
        c = self._caches.get(name)
        if c is None:
            c = self._caches[name] = _PackCache()
        return c.get(conv, algo)

    def _conv(self, name, conv, x, **kw):
        algo = pick_algo(conv.in_channels, conv.out_channels, 1, x.shape[-1], self.use_winograd,
                         plain=kw.get("residual") is None and not kw.get("pixel_shuffle"),
                         f4=self.use_winograd4)
        return conv3x3(x, self._packed(name, conv, algo), conv.out_channels, bias=conv.bias, algo=algo, **kw)

    def forward(self, x):
        """eval() mode or torch.no_grad(): the fused inference path.  train() mode with grad enabled (or an input that
        requires grad): the autograd path below."""
        L.require_gpu()
        if not x.is_cuda:
            raise L.KmsrError("SRNet.forward runs on the GPU only (no CPU fallback); move the input to cuda")
        if torch.is_grad_enabled() and (self.training or x.requires_grad):
            return self._forward_train(x)
        with torch.no_grad():
            return self._forward_infer(x)

    def _forward_train(self, x):
        """Same network with autograd: every conv is Conv3x3Function (HIP forward, dgrad and wgrad) with the residual adds
        and PixelShuffle fused into the forward kernels' epilogues; only the backward of PixelShuffle is a torch permutation."""
        cv = lambda conv, t, relu=False, residual=None, ps=False: Conv3x3Function.apply(t, conv.weight, conv.bias, relu, residual, ps)
        f = cv(self.head, x)
        h = f
        for blk in self.body:
            h = ResBlockFunction.apply(h, blk.conv1.weight, blk.conv1.bias, blk.conv2.weight, blk.conv2.bias)
        h = cv(self.body_tail, h, residual=f)
        ups = list(self.up)
        fuse = bool(ups) and self.fuse_up_tail_backward and smallci_unshuffle_eligible(self.n_colors, self.n_feats, 2 * h.shape[-2] * 2 ** (len(ups) - 1),
                                                                                       2 * h.shape[-1] * 2 ** (len(ups) - 1))
        for up in (ups[:-1] if fuse else ups):
            h = cv(up, h, ps=True)
        if fuse:        # last upsampler conv + tail as one node: the tail's data gradient is written un-shuffled (UpTailFunction)
            return UpTailFunction.apply(h, ups[-1].weight, ups[-1].bias, self.tail.weight, self.tail.bias)
        return cv(self.tail, h)

    def _forward_infer(self, x):
        f = self._conv("head", self.head, x)
        h = f
        for i, blk in enumerate(self.body):
            t = self._conv(f"b{i}.1", blk.conv1, h, relu=True)
            h = self._conv(f"b{i}.2", blk.conv2, t, residual=h)
        h = self._conv("body_tail", self.body_tail, h, residual=f)
        if self.fuse_upsampler and len(self.up) == 2 and self.n_colors <= _FusedUpsampler.MAX_BANDS \
                and min(h.shape[-2:]) >= _FusedUpsampler.MIN_SIZE \
                and h[0].numel() * 4 * min(h.shape[0], 64) < 2 ** 31:      # the frame launch's 32-bit offsets (kmsr_upsampler_frame_f32)
            return self._fused_up(self, h)
        for i, up in enumerate(self.up):
            h = self._conv(f"up{i}", up, h, pixel_shuffle=True)      # conv F->4F with PixelShuffle(2) fused into the store
        return self._conv("tail", self.tail, h)

    def flops_per_lr_pixel(self):
        F, C, B = self.n_feats, self.n_colors, len(self.body)
        per = 18 * (C * F + (2 * B + 1) * F * F)
        s = 1
        for _ in self.up:
            per += 18 * F * 4 * F * s * s
            s *= 2
        return per + 18 * F * C * s * s


def psnr(a, b, data_range):
    """10*log10(R^2 / MSE), fp64 reduction (the reference has no PSNR code; R must be declared)."""
    mse = torch.mean((a.double() - b.double()) ** 2)
    return float(10.0 * torch.log10(torch.tensor(float(data_range) ** 2, dtype=torch.float64) / mse))
