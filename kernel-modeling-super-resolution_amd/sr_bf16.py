"""bf16 inference path of the SR network (BASELINE config 4: 7-band x4 SR on 1024x1024 LR tiles, HBM-bound).

Activations are channels-last bf16 with the channel count padded to a multiple of 16 ([N,H,W,Cp] contiguous); the convs
run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (include/kmsr.h: kmsr_conv3x3_fwd_bf16).  The wrapper takes the
SAME SRNet module (fp32 master weights, standard state_dict) and the usual [B,C,h,w] tensor and returns [B,C,4h,4w].
"""
import torch

from . import _lib as L


def _pad16(c):
    return (c + 15) // 16 * 16


def to_nhwc_bf16(x, cp=None):
    """[N,C,H,W] (any float dtype) -> channels-last bf16 [N,H,W,Cp], zero padded channels."""
    N, C, H, W = x.shape
    cp = cp or _pad16(C)
    out = torch.zeros((N, H, W, cp), device=x.device, dtype=torch.bfloat16)
    out[..., :C] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    return out


def from_nhwc(x, c):
    """[N,H,W,Cp] -> [N,c,H,W] view-permuted (channels-last memory format)."""
    return x[..., :c].permute(0, 3, 1, 2)


def pack_conv3x3_weight_bf16(weight, cin_p, cout_p):
    lib = L.load()
    w = weight.detach().to(torch.float32).contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    packed = torch.empty(lib.kmsr_conv3x3_bf16_packed_weight_elems(cin_p, cout_p), device=w.device, dtype=torch.bfloat16)
    with torch.cuda.device(w.device):
        L.check(lib.kmsr_conv3x3_bf16_pack_weights(L.ptr(w), Cin, Cout, cin_p, cout_p, L.ptr(packed), L.current_stream()))
    return packed


def conv3x3_bf16(x, packed_w, cout_p, bias=None, relu=False, residual=None, pixel_shuffle=False, out=None):
    """x [N,H,W,Cin_p] bf16 contiguous -> [N,H,W,cout_p] (or [N,2H,2W,cout_p/4]); bias fp32 [>= cout_p] or None."""
    lib = L.load()
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.ndim == 4
    N, H, W, cin_p = x.shape
    d = L.Conv3x3Desc()
    d.N, d.Cin, d.H, d.W, d.Cout, d.stride = N, cin_p, H, W, int(cout_p), 1
    d.relu, d.pixel_shuffle = int(bool(relu)), int(bool(pixel_shuffle))
    assert packed_w.dtype == torch.bfloat16 and packed_w.numel() == lib.kmsr_conv3x3_bf16_packed_weight_elems(cin_p, int(cout_p))
    oshape = (N, 2 * H, 2 * W, cout_p // 4) if pixel_shuffle else (N, H, W, cout_p)
    if out is None:
        out = torch.empty(oshape, device=x.device, dtype=torch.bfloat16)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() >= cout_p
    if residual is not None:
        assert tuple(residual.shape) == oshape and residual.dtype == torch.bfloat16 and residual.is_contiguous()
    with torch.cuda.device(x.device):
        L.check(lib.kmsr_conv3x3_fwd_bf16(d, L.ptr(x), L.ptr(packed_w), L.ptr(bias), L.ptr(residual), L.ptr(out),
                                          L.current_stream()))
    return out


def resblock_bf16(x, packed_w1, bias1, packed_w2, bias2, out=None):
    """x [N,H,W,64] bf16 channels-last -> x + conv2(relu(conv1(x) + b1)) + b2 in ONE launch (kmsr_resblock_fwd_bf16): the intermediate
    map stays in LDS.  Bit-identical to conv3x3_bf16(relu=True) followed by conv3x3_bf16(residual=x)."""
    lib = L.load()
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.ndim == 4 and x.shape[3] == 64
    N, H, W, _ = x.shape
    if out is None:
        out = torch.empty_like(x)
    assert out.data_ptr() != x.data_ptr() and out.shape == x.shape and out.is_contiguous() and out.dtype == torch.bfloat16
    for pw in (packed_w1, packed_w2):
        assert pw.dtype == torch.bfloat16 and pw.numel() == lib.kmsr_conv3x3_bf16_packed_weight_elems(64, 64)
    for b in (bias1, bias2):
        assert b is None or (b.dtype == torch.float32 and b.numel() >= 64 and b.is_cuda)
    with torch.cuda.device(x.device):
        L.check(lib.kmsr_resblock_fwd_bf16(N, H, W, L.ptr(x), L.ptr(packed_w1), L.ptr(bias1), L.ptr(packed_w2), L.ptr(bias2), L.ptr(out),
                                           L.current_stream()))
    return out


def pack_conv3x3_weight_bf16_dgrad(weight, cin_p, cout_p):
    """Weights of the data-gradient conv (cout_p channels in, cin_p out): w'[ci,co,u,v] = w[co,ci,2-u,2-v], from the fp32 master."""
    lib = L.load()
    w = weight.detach().to(torch.float32).contiguous()
    Cout, Cin = w.shape[0], w.shape[1]
    packed = torch.empty(lib.kmsr_conv3x3_bf16_packed_weight_elems(cout_p, cin_p), device=w.device, dtype=torch.bfloat16)
    with torch.cuda.device(w.device):
        L.check(lib.kmsr_conv3x3_bf16_pack_weights_dgrad(L.ptr(w), Cin, Cout, cin_p, cout_p, L.ptr(packed), L.current_stream()))
    return packed


def _fwd_desc(x, cout_p):
    N, H, W, cin_p = x.shape
    d = L.Conv3x3Desc()
    d.N, d.Cin, d.H, d.W, d.Cout, d.stride = N, cin_p, H, W, int(cout_p), 1
    return d


def conv3x3_dgrad_bf16(dy, packed_wT, cin_p):
    """dy [N,H,W,cout_p] bf16 -> dx [N,H,W,cin_p] bf16  (kmsr_conv3x3_dgrad_bf16; packed_wT from pack_conv3x3_weight_bf16_dgrad)."""
    lib = L.load()
    assert dy.is_cuda and dy.dtype == torch.bfloat16 and dy.is_contiguous() and dy.ndim == 4
    N, H, W, cout_p = dy.shape
    d = L.Conv3x3Desc()
    d.N, d.Cin, d.H, d.W, d.Cout, d.stride = N, int(cin_p), H, W, cout_p, 1
    dx = torch.empty((N, H, W, cin_p), device=dy.device, dtype=torch.bfloat16)
    with torch.cuda.device(dy.device):
        L.check(lib.kmsr_conv3x3_dgrad_bf16(d, L.ptr(dy), L.ptr(packed_wT), L.ptr(dx), L.current_stream()))
    return dx


def conv3x3_wgrad_bf16(x, dy):
    """x [N,H,W,cin_p], dy [N,H,W,cout_p] bf16 -> dw fp32 [cout_p, cin_p, 3, 3]  (kmsr_conv3x3_wgrad_bf16)."""
    lib = L.load()
    assert x.is_cuda and x.dtype == dy.dtype == torch.bfloat16 and x.is_contiguous() and dy.is_contiguous()
    assert x.shape[:3] == dy.shape[:3]
    d = _fwd_desc(x, dy.shape[3])
    nbytes = lib.kmsr_conv3x3_wgrad_bf16_workspace_bytes(d)
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
    dw = torch.empty((dy.shape[3], x.shape[3], 3, 3), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        L.check(lib.kmsr_conv3x3_wgrad_bf16(d, L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(ws), nbytes, L.current_stream()))
    return dw


class Conv3x3BF16Function(torch.autograd.Function):
    """y = [relu](conv3x3(x, w) + b) on channels-last bf16 activations [N,H,W,Cp] with fp32 master weights [Cout,Cin,3,3] / bias:
    forward kmsr_conv3x3_fwd_bf16, backward kmsr_conv3x3_dgrad_bf16 + kmsr_conv3x3_wgrad_bf16 (fp32 weight / bias gradients)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        cin_p, cout_p = x.shape[3], _pad16(weight.shape[0])
        b = None
        if bias is not None:
            b = torch.zeros(max(cout_p, 64), device=x.device, dtype=torch.float32)
            b[:bias.numel()] = bias.detach().float()
        y = conv3x3_bf16(x.contiguous(), pack_conv3x3_weight_bf16(weight, cin_p, cout_p), cout_p, bias=b, relu=relu)
        ctx.relu, ctx.has_bias = relu, bias is not None
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        cout, cin = weight.shape[0], weight.shape[1]
        cin_p, cout_p = x.shape[3], _pad16(cout)
        g = g.contiguous()
        if ctx.relu:
            g = torch.where(y > 0, g, torch.zeros_like(g))
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = conv3x3_dgrad_bf16(g, pack_conv3x3_weight_bf16_dgrad(weight, cin_p, cout_p), cin_p)
        if ctx.needs_input_grad[1]:
            dw = conv3x3_wgrad_bf16(x, g)[:cout, :cin].contiguous()
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = g.float().sum(dim=(0, 1, 2))[:cout]
        return dx, dw, db, None


def conv5x5_ps4_bf16(x, packed_w, cout, bias=None, out=None):
    """x [N,H,W,Cin_p] bf16 channels-last -> bf16 planes [N, cout/16, 4H, 4W] (kmsr_conv5x5_ps4_fwd_bf16: the collapsed x4 upsampler)."""
    lib = L.load()
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.ndim == 4
    N, H, W, cin_p = x.shape
    assert packed_w.dtype == torch.bfloat16 and packed_w.numel() == lib.kmsr_conv5x5_ps4_bf16_packed_weight_elems(cin_p, int(cout))
    oshape = (N, cout // 16, 4 * H, 4 * W)
    if out is None:
        out = torch.empty(oshape, device=x.device, dtype=torch.bfloat16)
    else:
        assert tuple(out.shape) == oshape and out.is_contiguous() and out.dtype == torch.bfloat16
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == cout and bias.is_cuda
    with torch.cuda.device(x.device):
        L.check(lib.kmsr_conv5x5_ps4_fwd_bf16(N, cin_p, H, W, int(cout), L.ptr(x), L.ptr(packed_w), L.ptr(bias), L.ptr(out),
                                              L.current_stream()))
    return out


def pack_conv5x5_ps4_weight_bf16(w5, cin_p):
    """composed 5x5 weights [16*bands, F, 5, 5] fp32 -> the bf16 pack of kmsr_conv5x5_ps4_fwd_bf16."""
    lib = L.load()
    w = w5.detach().to(torch.float32).contiguous()
    cout, cin = w.shape[0], w.shape[1]
    packed = torch.empty(lib.kmsr_conv5x5_ps4_bf16_packed_weight_elems(cin_p, cout), device=w.device, dtype=torch.bfloat16)
    with torch.cuda.device(w.device):
        L.check(lib.kmsr_conv5x5_ps4_bf16_pack_weights(L.ptr(w), cin, cout, cin_p, L.ptr(packed), L.current_stream()))
    return packed


class SRNetBF16:
    """bf16 executor of an SRNet (weights are packed once from the module's fp32 parameters; call .refresh() after
    changing them).

    fuse_upsampler (default, x4 nets): up1 -> PS -> up2 -> PS -> tail (no non-linearity) runs as ONE bf16 conv5x5 + PixelShuffle(4)
    launch on the composed weights (sr.compose_upsampler, float64 on the host, rounded to bf16 once), writing the [N,bands,4h,4w]
    output directly; the outer 4-pixel HR frame -- where up2's and tail's own zero padding reach, so the composed operator does not
    apply -- is recomputed layer by layer on four 4-pixel-wide LR border strips and copied over the interior launch's output
    (bit-identical there to the layer-by-layer path).  The result is [N,C,4h,4w] bf16 planes; with fuse_upsampler = False (or other
    scales / tiny inputs) the three layers run one by one and the result is a channels-last view."""
    FUSE_MIN_SIZE = 8

    def __init__(self, net):
        self.net = net
        self.fuse_upsampler = True
        self.fuse_resblocks = True      # 64-feature nets: one kmsr_resblock_fwd_bf16 launch per residual block (bit-identical to two conv launches)
        self.refresh()

    def _prep(self, conv, cin_p, cout_p):
        dev = conv.weight.device
        b = torch.zeros(max(cout_p, 64), device=dev, dtype=torch.float32)
        b[:conv.out_channels] = conv.bias.detach().float()
        return pack_conv3x3_weight_bf16(conv.weight, cin_p, cout_p), b, cout_p

    @torch.no_grad()
    def refresh(self):
        n = self.net
        F = n.n_feats
        assert F % 64 == 0 or F in (16, 32, 48), "n_feats must suit the 16-channel MFMA k-step"
        self.cin_p = _pad16(n.n_colors)
        self.head = self._prep(n.head, self.cin_p, F)
        self.body = [(self._prep(b.conv1, F, F), self._prep(b.conv2, F, F)) for b in n.body]
        self.body_tail = self._prep(n.body_tail, F, F)
        self.up = [self._prep(u, F, 4 * F) for u in n.up]
        self.tail = self._prep(n.tail, F, _pad16(n.n_colors))
        self.up5 = None
        if len(n.up) == 2:
            from .sr import compose_upsampler
            w5, b5 = compose_upsampler(n.up[0], n.up[1], n.tail)
            dev = n.tail.weight.device
            self.up5 = (pack_conv5x5_ps4_weight_bf16(w5.to(dev), F), b5.to(dev).contiguous(), w5.shape[0])

    @torch.no_grad()
    def __call__(self, x):
        L.require_gpu()
        n = self.net
        h0 = to_nhwc_bf16(x, self.cin_p)
        cv = lambda prm, t, **kw: conv3x3_bf16(t, prm[0], prm[2], bias=prm[1], **kw)
        f = cv(self.head, h0)
        h = f
        if self.fuse_resblocks and n.n_feats == 64:
            pp = [torch.empty_like(h), torch.empty_like(h)]              # ping-pong (the fused kernel is not in-place; f stays intact)
            for i, (c1, c2) in enumerate(self.body):
                h = resblock_bf16(h, c1[0], c1[1], c2[0], c2[1], out=pp[i & 1])
        else:
            for c1, c2 in self.body:
                h = cv(c2, cv(c1, h, relu=True), residual=h)
        h = cv(self.body_tail, h, residual=f)
        if self.fuse_upsampler and self.up5 is not None and min(h.shape[1:3]) >= self.FUSE_MIN_SIZE:
            return self._fused_upsampler(h)
        return from_nhwc(self._layerwise_upsampler(h), n.n_colors)

    def _layerwise_upsampler(self, h):
        cv = lambda prm, t, **kw: conv3x3_bf16(t, prm[0], prm[2], bias=prm[1], **kw)
        for u in self.up:
            h = cv(u, h, pixel_shuffle=True)
        return cv(self.tail, h)

    def _fused_upsampler(self, h):
        """h [N,H,W,F] bf16 channels-last -> [N,bands,4H,4W] bf16 planes."""
        C = self.net.n_colors
        out = conv5x5_ps4_bf16(h, self.up5[0], self.up5[2], bias=self.up5[1])
        # the outer HR frame (one LR row / column = 4 HR pixels) from the layers themselves: a 4-wide LR strip holds everything
        # the frame's pixels depend on (HR rows 0..3 <- 4x rows 0..4 <- 2x rows 0..3 <- LR rows 0..2)
        N, H, W = h.shape[0], h.shape[1], h.shape[2]
        tb = self._layerwise_upsampler(torch.cat((h[:, :4], h[:, H - 4:]), dim=0))            # top and bottom strips as one batch: [2N,16,4W,Cp]
        lr = self._layerwise_upsampler(torch.cat((h[:, :, :4], h[:, :, W - 4:]), dim=0))      # left and right: [2N,4H,16,Cp]
        out[:, :, :4, :] = tb[:N, :4, :, :C].permute(0, 3, 1, 2)
        out[:, :, 4 * H - 4:, :] = tb[N:, 12:, :, :C].permute(0, 3, 1, 2)
        out[:, :, :, :4] = lr[:N, :, :4, :C].permute(0, 3, 1, 2)
        out[:, :, :, 4 * W - 4:] = lr[N:, :, 12:, :C].permute(0, 3, 1, 2)
        return out
