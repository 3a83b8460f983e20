"""CPU-side checks of the drop-in boundary: libkmsr_hip.so loads and exports every symbol that
include/kmsr.h declares; host-only entry points behave; the product package refuses to run without
a GPU instead of silently falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import kmsr_amd
from kmsr_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "kmsr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kmsr_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported_and_bound():
    lib = L.load()
    names = _declared_symbols()
    assert "kmsr_degrade_f32" in names and "kmsr_version" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/kmsr.h but not exported by libkmsr_hip.so"
        assert n in L.SYMBOLS, f"{n} has no ctypes prototype in _lib.SYMBOLS"
    assert sorted(L.SYMBOLS) == names, "ctypes table and header disagree"
    assert lib.kmsr_version() == 101


def test_struct_layout_matches_header():
    # kmsr_degrade_desc: 4*i32, 3*i64, 9*i32 + u32, u64, u32, i32, u32, u32  (natural alignment)
    assert C.sizeof(L.DegradeDesc) == 16 + 24 + 40 + 8 + 8 + 8
    assert L.DegradeDesc.hr_stride_n.offset == 16 and L.DegradeDesc.seed.offset == 80
    assert L.DegradeDesc.noise_sample_base.offset == 96 and L.DegradeDesc.noise_sample_step.offset == 100


_CTYPE_OF = {"int32_t": C.c_int32, "uint32_t": C.c_uint32, "int64_t": C.c_int64, "uint64_t": C.c_uint64}


def _header_struct_fields(name):
    """(field, ctype) list of `typedef struct <name> {...}` as include/kmsr.h declares it."""
    src = open(os.path.join(ROOT, "include", "kmsr.h")).read()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for ty, names in re.findall(r"\b(u?int(?:32|64)_t)\s+([^;]+);", body):
        out += [(n.strip(), _CTYPE_OF[ty]) for n in names.split(",")]
    return out


def test_integration_md_binding_matches_header_and_lib():
    """The ctypes stub INTEGRATION.md section B tells a maintainer to copy must be the struct the library reads
    (round 3: the document lagged the header by two fields)."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    snippet = re.search(r"class DegradeDesc\(C\.Structure\):.*?_fields_ = (\[.*?\])\s*(?:#[^\n]*)?\n\n", doc, flags=re.S).group(1)
    snippet = re.sub(r"#[^\n]*", "", snippet)
    doc_fields = eval(snippet, {"C": C})                       # a list literal of (name, C.c_*) pairs, our own document
    lib_fields = [(n, t) for n, t in L.DegradeDesc._fields_]
    assert doc_fields == lib_fields, "INTEGRATION.md DegradeDesc differs from _lib.DegradeDesc"
    assert lib_fields == _header_struct_fields("kmsr_degrade_desc"), "_lib.DegradeDesc differs from include/kmsr.h"

    class Doc(C.Structure):
        _fields_ = doc_fields
    assert C.sizeof(Doc) == C.sizeof(L.DegradeDesc) == 104
    assert [(n, t) for n, t in L.Conv3x3Desc._fields_] == _header_struct_fields("kmsr_conv3x3_desc")


def _desc(**kw):
    d = L.DegradeDesc()
    d.N, d.C, d.H, d.W = 1, 5, 512, 512
    d.hr_stride_n, d.hr_stride_c, d.hr_stride_h = 5 * 512 * 512, 512 * 512, 512
    d.n_kernels, d.kh, d.kw, d.factor = 1, 13, 13, 8
    for k, v in kw.items():
        setattr(d, k, v)
    return d


@pytest.mark.parametrize("H,W,f,mode,want", [
    (512, 512, 8, L.DOWN_AVGPOOL, (64, 64)), (512, 512, 4, L.DOWN_AVGPOOL, (128, 128)),
    (96, 96, 6, L.DOWN_AVGPOOL, (24, 24)),            # int(log2(6)) = 2 -> x4   (C_30:120)
    (70, 90, 8, L.DOWN_AVGPOOL, (8, 11)),             # floor at every pooling level
    (64, 64, 1, L.DOWN_AVGPOOL, (64, 64)), (7, 7, 8, L.DOWN_AVGPOOL, (0, 0)),
    (250, 253, 4, L.DOWN_STRIDE, (63, 64)),           # [::4] -> ceil     (train_gemini.py:134)
    (256, 256, 3, L.DOWN_STRIDE, (86, 86)),
])
def test_out_shape_host_arithmetic(H, W, f, mode, want):
    lib = L.load()
    h, w = C.c_int32(), C.c_int32()
    assert lib.kmsr_degrade_out_shape(_desc(H=H, W=W, factor=f, down_mode=mode), h, w) == 0
    assert (h.value, w.value) == want
    # the reference's own formula
    if mode == L.DOWN_AVGPOOL:
        hh, ww = H, W
        for _ in range(int(np.log2(f))):
            hh, ww = hh // 2, ww // 2
        assert (hh, ww) == want


def test_error_codes_and_messages_without_gpu():
    lib = L.load()
    h, w = C.c_int32(), C.c_int32()
    assert lib.kmsr_degrade_out_shape(_desc(kh=12), h, w) == L.KMSR_E_UNSUPPORTED
    assert b"kernel size" in lib.kmsr_last_error()
    assert lib.kmsr_degrade_out_shape(_desc(factor=0), h, w) == L.KMSR_E_BADARG
    assert lib.kmsr_degrade_out_shape(_desc(down_mode=7), h, w) == L.KMSR_E_BADARG
    assert lib.kmsr_degrade_out_shape(None, h, w) == L.KMSR_E_BADARG
    with pytest.raises(ValueError):
        L.check(L.KMSR_E_BADARG)
    with pytest.raises(AssertionError):
        L.check(L.KMSR_E_SHAPE)
    with pytest.raises(MemoryError):
        L.check(L.KMSR_E_NOMEM)
    # null tensors are rejected before any launch
    d = _desc()
    assert lib.kmsr_degrade_f32(d, None, None, None, None, None, None, None, None, 0, None) == L.KMSR_E_BADARG
    # the effective tap table [rows][TY][TXP] fp32 and its row-reversed copy (read by the bottom-up chunks of the row-streaming kernel)
    assert lib.kmsr_degrade_workspace_bytes(_desc(factor=4)) == 2 * 5 * 16 * 16 * 4
    assert lib.kmsr_degrade_workspace_bytes(_desc(factor=4, down_mode=L.DOWN_STRIDE)) == 2 * 5 * 13 * 16 * 4


def test_upsampler_frame_argument_checks_without_gpu():
    """kmsr_upsampler_frame_f32 rejects what its MFMA kernel cannot address BEFORE any launch (fake pointers are never dereferenced):
    misaligned class biases / output (16-byte accesses), and min(N, 64) input images beyond the 2 GB a workgroup's buffer descriptor spans
    (SRNet then runs the upsampler layer by layer: sr.py gate)."""
    lib = L.load()
    p = lambda a: C.c_void_p(a)
    call = lambda h, N, F, H, W, b, wc, bc, y: lib.kmsr_upsampler_frame_f32(p(h), N, F, H, W, b, p(wc), p(bc), p(y), None)
    assert call(0, 1, 64, 16, 16, 3, 0x1000, 0x1000, 0x1000) == L.KMSR_E_BADARG            # null input
    assert call(0x1000, 1, 64, 16, 16, 9, 0x1000, 0x1000, 0x1000) == L.KMSR_E_UNSUPPORTED   # 9 bands
    assert call(0x1000, 1, 64, 16, 16, 3, 0x1000, 0x1004, 0x1000) == L.KMSR_E_BADARG        # class biases not 16-byte aligned
    assert call(0x1000, 1, 64, 16, 16, 3, 0x1000, 0x1000, 0x1008) == L.KMSR_E_BADARG        # output not 16-byte aligned
    assert call(0x1000, 64, 64, 512, 512, 3, 0x1000, 0x1000, 0x1000) == L.KMSR_E_UNSUPPORTED
    assert b"2 GB" in lib.kmsr_last_error()


def test_wino4_argument_checks_without_gpu():
    """kmsr_conv3x3_wino4_fwd_f32 rejects Cout beyond its LDS bias table with AND without a bias pointer (the epilogue reads the
    table either way: round-3 advisor finding), plus the alignment / width rules, before any launch."""
    lib = L.load()
    d = L.Conv3x3Desc()
    d.N, d.Cin, d.H, d.W, d.Cout, d.stride = 1, 64, 16, 16, 2048, 1
    p = lambda a: C.c_void_p(a)
    assert lib.kmsr_conv3x3_wino4_fwd_f32(d, p(0x1000), p(0x1000), None, None, p(0x1000), None) == L.KMSR_E_UNSUPPORTED
    assert b"Cout <=" in lib.kmsr_last_error()
    assert lib.kmsr_conv3x3_wino4_fwd_f32(d, p(0x1000), p(0x1000), p(0x1000), None, p(0x1000), None) == L.KMSR_E_UNSUPPORTED
    d.Cout, d.W = 64, 18
    assert lib.kmsr_conv3x3_wino4_fwd_f32(d, p(0x1000), p(0x1000), None, None, p(0x1000), None) == L.KMSR_E_UNSUPPORTED
    d.W = 16
    assert lib.kmsr_conv3x3_wino4_fwd_f32(d, p(0x1004), p(0x1000), None, None, p(0x1000), None) == L.KMSR_E_BADARG


def test_direct_conv_argument_checks_without_gpu():
    """kmsr_conv3x3_fwd_f32 (round 4: input tiles staged by buffer loads with 32-bit byte offsets inside an 8-channel chunk) rejects planes whose
    chunk does not fit those offsets, and the shapes it never served, before any launch."""
    lib = L.load()
    d = L.Conv3x3Desc()
    p = lambda a: C.c_void_p(a)
    d.N, d.Cin, d.H, d.W, d.Cout, d.stride = 1, 8, 16384, 8192, 64, 1                   # 2^27 pixels per plane: 8 planes = 2^32 bytes
    assert lib.kmsr_conv3x3_fwd_f32(d, p(0x1000), p(0x1000), None, None, p(0x1000), None) == L.KMSR_E_UNSUPPORTED
    assert b"32-bit buffer offsets" in lib.kmsr_last_error()
    d.Cout = 3                                                                           # the small-Cout kernel stages the same way
    assert lib.kmsr_conv3x3_fwd_f32(d, p(0x1000), p(0x1000), None, None, p(0x1000), None) == L.KMSR_E_UNSUPPORTED
    d.H, d.W, d.Cout, d.stride = 16, 16, 64, 3
    assert lib.kmsr_conv3x3_fwd_f32(d, p(0x1000), p(0x1000), None, None, p(0x1000), None) == L.KMSR_E_UNSUPPORTED
    d.stride = 1
    assert lib.kmsr_conv3x3_fwd_f32(d, None, p(0x1000), None, None, p(0x1000), None) == L.KMSR_E_BADARG


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(L.KmsrError):
        kmsr_amd.apply_kernel_degradation(torch.zeros(5, 64, 64), torch.ones(5, 13, 13), 8)
    with pytest.raises(L.KmsrError):
        kmsr_amd.add_noise(np.zeros((5, 32, 32), np.float32), np.zeros((4, 5, 32, 32), np.float32))


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "kernel-modeling-super-resolution_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M), f"{f} imports the oracle"
                assert "libkmsr_oracle" not in txt, f"{f} references the oracle library"


def test_load_kernel_semantics(tmp_path):
    """C_31:22-37 against the fixture produced by the reference's own load_kernel."""
    from conftest import golden
    lk = golden("load_kernel.npz")
    p = str(tmp_path / "k.npy")
    np.save(p, lk["batch_in"])
    got = kmsr_amd.load_kernel(p, verbose=False)
    assert got.dtype.is_floating_point and tuple(got.shape) == (5, 13, 13)
    np.testing.assert_allclose(got.numpy(), lk["batch_out"], rtol=1e-6)
    np.save(p, lk["k2d_in"])
    assert np.array_equal(kmsr_amd.load_kernel(p, verbose=False).numpy(), lk["k2d_out_c31"])


def test_random_crop_matches_reference_stream():
    """D_build_noise_pool.py:41-53"""
    import random
    from conftest import golden
    rng = golden("rng_streams.npz")
    data = np.arange(5 * 256 * 256, dtype=np.float32).reshape(5, 256, 256)
    random.seed(42)
    crops = kmsr_amd.random_crop(data, 32, 16)
    offs = [(int(c[0, 0, 0]) // 256, int(c[0, 0, 0]) % 256) for c in crops]
    assert offs == [tuple(o) for o in rng["random_crop_offs_seed42_256"]]
    assert all(c.shape == (5, 32, 32) for c in crops)
    with pytest.raises(ValueError):
        kmsr_amd.random_crop(data[:, :16], 32, 1)


def test_compose_upsampler_is_the_interior_operator():
    """Host-side check (no GPU call): the composed 5x5 weights reproduce up1/PS/up2/PS/tail from HR pixel 3 inwards."""
    import torch
    import torch.nn.functional as F
    from kmsr_amd.sr import SRNet, compose_upsampler
    torch.manual_seed(3)
    net = SRNet(3, 16, 1, 4)
    w5, b5 = compose_upsampler(net.up[0], net.up[1], net.tail)
    h = torch.randn(2, 16, 12, 14, dtype=torch.float64)
    r = h
    for cv in net.up:
        r = F.pixel_shuffle(F.conv2d(r, cv.weight.detach().double(), cv.bias.detach().double(), padding=1), 2)
    r = F.conv2d(r, net.tail.weight.detach().double(), net.tail.bias.detach().double(), padding=1)
    f = F.pixel_shuffle(F.conv2d(h, w5.double(), b5.double(), padding=2), 4)
    d = (r - f).abs()
    assert float(d[:, :, 3:-3, 3:-3].max()) < 1e-6 and float(d.max()) > 1e-3   # the frame is where they differ


def test_pick_algo_dispatch():
    """Host-side kernel choice for a stride-1 3x3 conv (kmsr_amd/sr.py: pick_algo)."""
    from kmsr_amd.sr import pick_algo
    assert pick_algo(64, 64, 1, 256) == "winograd_vreg"         # rows 16-byte aligned -> persistent LDS-DMA Winograd kernel
    assert pick_algo(64, 256, 1, 512) == "winograd_vreg"
    assert pick_algo(64, 64, 1, 70) == "winograd"               # even width only -> register-staged Winograd
    assert pick_algo(64, 64, 1, 71) == "direct"                 # odd width
    assert pick_algo(3, 64, 1, 256) == "direct"                 # head: too few input channels
    assert pick_algo(64, 3, 1, 1024) == "direct"                # tail: small-Cout VALU kernel
    assert pick_algo(64, 64, 2, 256) == "direct"                # stride 2 (selector)
    assert pick_algo(64, 64, 1, 256, use_winograd=False) == "direct"


def test_pick_algo_routes_layers_to_their_kernels():
    """Host-side dispatch of the 3x3 layers (no GPU needed): which C-ABI entry point serves which layer shape."""
    from kmsr_amd.sr import pick_algo
    assert pick_algo(64, 64, 1, 256) == "winograd_vreg"              # SR body: persistent V-in-registers Winograd kernel (W % 4 == 0)
    assert pick_algo(64, 256, 1, 254) == "winograd"                  # even width that is not a multiple of 4: register-staged Winograd
    assert pick_algo(64, 64, 1, 255) == "direct"                     # odd width
    assert pick_algo(64, 64, 2, 256) == "direct"                     # stride 2 (SelectorNet / ConditionEncoder)
    assert pick_algo(64, 5, 1, 1024) == "direct"                     # SR tail: Cout <= 8 -> the small-Cout VALU kernel behind kmsr_conv3x3_fwd_f32
    assert pick_algo(3, 64, 1, 256) == "direct"                      # head with an epilogue the small-Cin kernel does not have
    assert pick_algo(3, 64, 1, 256, plain=True) == "smallci"         # head / tail data gradient: bias + ReLU epilogue only
    assert pick_algo(5, 64, 1, 250, plain=True) == "direct"          # small-Cin kernel needs W % 4 == 0
    assert pick_algo(64, 64, 1, 256, use_winograd=False) == "direct"
    assert pick_algo(64, 2048, 1, 256) == "winograd"                 # bias vector would not fit the persistent kernel's LDS staging
    assert pick_algo(64, 64, 1, 256, f4=True) == "winograd4"         # SR body at inference (no PixelShuffle): Winograd F(4x4,3x3)
    assert pick_algo(64, 64, 1, 254, f4=True) == "winograd"          # ... needs 16-byte aligned rows like the F(2x2) persistent kernel
    assert pick_algo(3, 64, 1, 256, plain=True, f4=True) == "smallci"


def test_no_wide_store_with_sgpr_soffset():
    """Round-2's misplaced-dword bug (profiles/r03_soffset_store_hazard.log): on gfx950 a 12- / 16-byte MUBUF store that carries an SGPR
    soffset needs a wait state before its data VGPRs are overwritten, and hipcc (LLVM's gfx9 hazard table exempts that form) does not
    insert one.  No kernel may contain the form: compile the kernels that use buffer stores to ISA and grep."""
    import re
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "kernel-modeling-super-resolution_amd", "csrc")
    pat = re.compile(r"buffer_store_dwordx[34]\s+v\[\d+:\d+\],\s*v\d+,\s*s\[\d+:\d+\],\s*s\d+")
    with tempfile.TemporaryDirectory() as td:
        for f in ("conv3x3_wino_vreg.hip", "conv3x3_wino4.hip", "conv3x3_smallci.hip"):
            out = os.path.join(td, f + ".s")
            subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-fno-slp-vectorize", "-w", "-S",
                                   "--cuda-device-only", "-o", out, os.path.join(csrc, f)], stderr=subprocess.DEVNULL)
            isa = open(out).read()
            assert "buffer_store_dword" in isa, f
            bad = pat.findall(isa)
            assert not bad, f"{f}: wide buffer store with an SGPR soffset: {bad[:2]}"
            if f == "conv3x3_wino4.hip":
                # round 3: a VGPR spilled across the F(4x4) kernel's steps is reloaded from scratch inside the epilogue, and every such
                # reload is an s_waitcnt vmcnt(0) on the stores and prefetches in flight (8 us per launch); the kernel must stay spill-free
                assert "scratch_" not in isa, "conv3x3_wino4.hip: VGPR spills to scratch (see DESIGN.md 4.4c)"


def test_resblock_mfma_results_are_not_read_within_11_wait_states():
    """Round 4: a VALU read of a v_mfma_f32_32x32x16_bf16 result is not interlocked on gfx950 -- it needs 11 wait
    states (tools/ubench/mfma_bf16_hazards.hip; profiles/r04_ubench_mfma_bf16_hazards.log), and hipcc inserts them for the builtin only.  The fused
    res-block kernel writes its MFMAs as inline asm, so the kernel itself has to keep every non-MFMA vector instruction that touches an MFMA's
    destination registers at least 11 wait states behind it (RB_DRAIN).  Compile the kernel and check exactly that on the ISA: every instruction
    counts one wait state, `s_nop N` counts N + 1 (other MFMAs in between occupy the pipe for longer than that: counting them as 1 is conservative)."""
    import re
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "kernel-modeling-super-resolution_amd", "csrc")

    def regs(text):
        out = set()
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else {int(m.group(3))}
        return out

    with tempfile.TemporaryDirectory() as td:
        # (v_mfma_f32_16x16x4_f32 needs no such distance -- measured interlocked by the same microbenchmark -- so the F(4x4) weight-gradient kernel,
        #  whose MFMAs are inline asm too, is only checked for the no-spill property)
        for f, mfma in (("resblock_bf16.hip", "v_mfma_f32_32x32x16_bf16"), ("conv3x3_wgrad_wino4.hip", None)):
            out = os.path.join(td, f + ".s")
            subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-fno-slp-vectorize", "-w", "-S",
                                   "--cuda-device-only", "-o", out, os.path.join(csrc, f)], stderr=subprocess.DEVNULL)
            if mfma is None:
                continue
            lines = [ln.strip() for ln in open(out) if ln.strip() and not ln.strip().startswith((";", "."))]
            n_mfma, bad = 0, []
            for i, ln in enumerate(lines):
                if not ln.startswith(mfma):
                    continue
                n_mfma += 1
                dst = regs(ln.split(None, 1)[1].split(",")[0])
                assert len(dst) == 16, ln                       # accumulators of this kernel live in VGPRs
                waited = 0
                for nxt in lines[i + 1:i + 16]:
                    if waited >= 11:
                        break
                    if nxt.startswith("s_nop"):
                        waited += int(nxt.split()[1]) + 1
                        continue
                    waited += 1
                    if nxt.startswith("v_mfma") or not nxt.startswith(("v_", "ds_", "buffer_", "global_")):
                        continue                                # dependent MFMAs are interlocked; scalar instructions cannot touch VGPRs
                    if len(nxt.split(None, 1)) > 1 and regs(nxt.split(None, 1)[1]) & dst:
                        bad.append((ln[:70], nxt[:70], waited))
            assert n_mfma >= 144, (f, n_mfma)
            assert not bad, f"{f}: MFMA result touched {bad[0][2]} wait states behind the MFMA: {bad[:2]}"
        # both kernels must stay free of VGPR spills (their accumulators / weights live in fixed register classes: DESIGN 4.5d, 4.7c)
        assert "scratch_" not in open(os.path.join(td, "conv3x3_wgrad_wino4.hip.s")).read(), "conv3x3_wgrad_wino4.hip spills VGPRs"
        assert "scratch_" not in open(os.path.join(td, "resblock_bf16.hip.s")).read(), "resblock_bf16.hip spills VGPRs"


def test_resblock_and_wgrad_argument_checks_without_gpu():
    """The round-4 entry points reject what their kernels cannot address before any launch (fake pointers are never dereferenced)."""
    lib = L.load()
    p = lambda a: C.c_void_p(a)
    rb = lambda N, H, W, x, w1, w2, out: lib.kmsr_resblock_fwd_bf16(N, H, W, p(x), p(w1), None, p(w2), None, p(out), None)
    assert rb(1, 8, 16, 0, 0x1000, 0x1000, 0x2000) == L.KMSR_E_BADARG                 # null input
    assert rb(1, 8, 16, 0x1000, 0x1000, 0x1000, 0x1000) == L.KMSR_E_BADARG            # in place
    assert b"in-place" in lib.kmsr_last_error()
    assert rb(1, 8, 16, 0x1008, 0x1000, 0x1000, 0x2000) == L.KMSR_E_BADARG            # misaligned
    assert rb(0, 8, 16, 0x1000, 0x1000, 0x1000, 0x2000) == L.KMSR_E_BADARG
    assert rb(1, 8192, 8192, 0x1000, 0x1000, 0x1000, 0x2000) == L.KMSR_E_UNSUPPORTED  # 8 GiB per image: 32-bit buffer offsets
    d = L.Conv3x3Desc()
    d.N, d.Cin, d.H, d.W, d.Cout, d.stride = 32, 64, 256, 256, 64, 1
    w4 = lib.kmsr_conv3x3_wgrad_workspace_bytes(d)                                    # F(4x4) domain: (nblk + 1) slabs of 36 x 64 x 32 floats per group
    assert w4 % (36 * 64 * 32 * 4) == 0 and w4 >= 2 * 2 * 36 * 64 * 32 * 4
    d.flags = L.CONV_FLAG_WGRAD_WINO2
    w2 = lib.kmsr_conv3x3_wgrad_workspace_bytes(d)                                    # F(2x2): slabs of 16 x 64 x 64
    assert w2 % (16 * 64 * 64 * 4) == 0 and w2 != w4
    d.flags = 8
    assert lib.kmsr_conv3x3_wgrad_f32(d, p(0x1000), p(0x1000), p(0x1000), p(0x1000), C.c_size_t(1 << 40), None) == L.KMSR_E_BADARG   # unknown flag
    d.flags = 0
    assert lib.kmsr_conv3x3_wgrad_f32(d, p(0x1004), p(0x1000), p(0x1000), p(0x1000), C.c_size_t(1 << 40), None) == L.KMSR_E_BADARG   # F(4x4): x not 16-byte aligned
