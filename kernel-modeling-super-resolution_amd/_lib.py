"""ctypes binding of libkmsr_hip.so (the C ABI declared in include/kmsr.h).

PyTorch is used for device memory and streams only: every tensor crosses the boundary as a raw
device pointer.  There is NO CPU fallback -- if the HIP library is missing or no GPU is visible the
product path raises (the CPU oracle under oracle/ is test infrastructure and is never imported here).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KMSR_LIB_OVERRIDE") or os.path.join(_HERE, "libkmsr_hip.so")   # override: diagnostic builds only

KMSR_OK = 0
KMSR_E_BADARG, KMSR_E_SHAPE, KMSR_E_HIP, KMSR_E_RCCL, KMSR_E_NOMEM, KMSR_E_UNSUPPORTED = -1, -2, -3, -4, -5, -6
DOWN_AVGPOOL, DOWN_STRIDE = 0, 1
PAD_REPLICATE, PAD_ZERO = 0, 1
NOISE_NONE, NOISE_GAUSSIAN, NOISE_POOL = 0, 1, 2
FLAG_FORCE_GENERIC, FLAG_FORCE_FAST, FLAG_TAPS_READY = 1, 2, 4


class KmsrError(RuntimeError):
    pass


class DegradeDesc(C.Structure):
    """struct kmsr_degrade_desc (include/kmsr.h)"""
    _fields_ = [
        ("N", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("hr_stride_n", C.c_int64), ("hr_stride_c", C.c_int64), ("hr_stride_h", C.c_int64),
        ("n_kernels", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
        ("factor", C.c_int32), ("down_mode", C.c_int32), ("pad_mode", C.c_int32),
        ("normalize", C.c_int32), ("noise_mode", C.c_int32), ("pool_size", C.c_int32),
        ("flags", C.c_uint32), ("seed", C.c_uint64), ("offset", C.c_uint32),
        ("rows_per_chunk", C.c_int32), ("noise_sample_base", C.c_uint32), ("noise_sample_step", C.c_uint32),
    ]


class Conv3x3Desc(C.Structure):
    """struct kmsr_conv3x3_desc (include/kmsr.h)"""
    _fields_ = [
        ("N", C.c_int32), ("Cin", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("Cout", C.c_int32), ("stride", C.c_int32), ("relu", C.c_int32), ("pixel_shuffle", C.c_int32),
        ("flags", C.c_uint32),
    ]


class ConvKDesc(C.Structure):
    """struct kmsr_convk_desc (include/kmsr.h)"""
    _fields_ = [
        ("U", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("k", C.c_int32),
        ("pad_mode", C.c_int32), ("shared_weights", C.c_int32),
    ]


CONVK_REFLECT, CONVK_ZERO = 0, 1
CONV_FLAG_RELU_MASK = 1
CONV_FLAG_WGRAD_DIRECT = 2
CONV_FLAG_WGRAD_WINO2 = 4

_lib = None

# every symbol include/kmsr.h declares: (name, restype, argtypes)
_vp, _sz, _i32, _u32, _u64 = C.c_void_p, C.c_size_t, C.c_int32, C.c_uint32, C.c_uint64
SYMBOLS = {
    "kmsr_version": (C.c_int, []),
    "kmsr_last_error": (C.c_char_p, []),
    "kmsr_device_count": (C.c_int, []),
    "kmsr_degrade_out_shape": (C.c_int, [C.POINTER(DegradeDesc), C.POINTER(_i32), C.POINTER(_i32)]),
    "kmsr_degrade_workspace_bytes": (_sz, [C.POINTER(DegradeDesc)]),
    "kmsr_degrade_uses_fast_path": (C.c_int, [C.POINTER(DegradeDesc), _vp]),
    "kmsr_degrade_f32": (C.c_int, [C.POINTER(DegradeDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kmsr_degrade_bwd_workspace_bytes": (_sz, [C.POINTER(DegradeDesc)]),
    "kmsr_degrade_bwd_f32": (C.c_int, [C.POINTER(DegradeDesc), _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kmsr_gaussian_eps_f32": (C.c_int, [_vp, _i32, _i32, _i32, _u64, _u32, _u32, _vp]),
    "kmsr_noise_crops_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp]),
    "kmsr_conv3x3_packed_weight_floats": (_sz, [_i32, _i32]),
    "kmsr_conv3x3_pack_weights_f32": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "kmsr_conv3x3_fwd_f32": (C.c_int, [C.POINTER(Conv3x3Desc), _vp, _vp, _vp, _vp, _vp, _vp]),
    "kmsr_pixel_unshuffle2_sum_partials": (_sz, [_i32, _i32, _i32]),
    "kmsr_pixel_unshuffle2_sum_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "kmsr_l1_loss_workspace_floats": (C.c_size_t, []),
    "kmsr_l1_loss_f32": (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp, _vp, _vp]),
    "kmsr_scale_by_device_scalar_f32": (C.c_int, [_vp, C.c_int64, _vp, _vp]),
    "kmsr_conv3x3_dgrad_f32": (C.c_int, [C.POINTER(Conv3x3Desc), _vp, _vp, _vp, _vp]),
    "kmsr_conv3x3_pixelshuffle_fwd_f32": (C.c_int, [C.POINTER(Conv3x3Desc), _vp, _vp, _vp, _vp, _vp]),
    "kmsr_conv3x3_pixelshuffle_fwd_bf16": (C.c_int, [C.POINTER(Conv3x3Desc), _vp, _vp, _vp, _vp, _vp]),
    "kmsr_conv3x3_smallci_fwd_f32": (C.c_int, [C.POINTER(Conv3x3Desc), _vp, _vp, _vp, _vp, _vp]),
    "kmsr_conv3x3_smallci_unshuffle_workspace_bytes": (_sz, [C.POINTER(Conv3x3Desc)]),
    "kmsr_conv3x3_smallci_unshuffle_fwd_f32": (C.c_int, [C.POINTER(Conv3x3Desc), _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "kmsr_conv3x3_wgrad_workspace_bytes": (_sz, [C.POINTER(Conv3x3Desc)]),
    "kmsr_conv3x3_wgrad_f32": (C.c_int, [C.POINTER(Conv3x3Desc), _vp, _vp, _vp, _vp, _sz, _vp]),
    "kmsr_conv3x3_wino_packed_weight_floats": (_sz, [_i32, _i32]),
    "kmsr_conv3x3_wino_pack_weights_f32": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "kmsr_conv3x3_winov_chan_sum_floats": (_sz, [C.POINTER(Conv3x3Desc)]),
    "kmsr_conv3x3_winov_fwd_ex_f32": (C.c_int, [C.POINTER(Conv3x3Desc), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "kmsr_conv3x3_wino_fwd_f32": (C.c_int, [C.POINTER(Conv3x3Desc), _vp, _vp, _vp, _vp, _vp, _vp]),
    "kmsr_conv3x3_bf16_packed_weight_elems": (_sz, [_i32, _i32]),
    "kmsr_conv3x3_bf16_pack_weights": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "kmsr_conv3x3_fwd_bf16": (C.c_int, [C.POINTER(Conv3x3Desc), _vp, _vp, _vp, _vp, _vp, _vp]),
    "kmsr_resblock_fwd_bf16": (C.c_int, [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "kmsr_conv3x3_bf16_pack_weights_dgrad": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "kmsr_conv3x3_dgrad_bf16": (C.c_int, [C.POINTER(Conv3x3Desc), _vp, _vp, _vp, _vp]),
    "kmsr_conv3x3_wgrad_bf16_workspace_bytes": (_sz, [C.POINTER(Conv3x3Desc)]),
    "kmsr_conv3x3_wgrad_bf16": (C.c_int, [C.POINTER(Conv3x3Desc), _vp, _vp, _vp, _vp, _sz, _vp]),
    "kmsr_water_mask_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, C.c_float, C.c_float, C.c_float, _vp, _vp, _vp]),
    "kmsr_patch_nan_counts_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "kmsr_gather_patches_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp]),
    "kmsr_nlm_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "kmsr_conv3x3_winov_packed_weight_floats": (_sz, [_i32, _i32]),
    "kmsr_conv3x3_winov_pack_weights_f32": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "kmsr_conv3x3_winov_fwd_f32": (C.c_int, [C.POINTER(Conv3x3Desc), _vp, _vp, _vp, _vp, _vp, _vp]),
    "kmsr_conv3x3_wino4_packed_weight_floats": (_sz, [_i32, _i32]),
    "kmsr_conv3x3_wino4_pack_weights_f32": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "kmsr_conv3x3_wino4_pack_weights_ex_f32": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "kmsr_conv3x3_wino4_fwd_f32": (C.c_int, [C.POINTER(Conv3x3Desc), _vp, _vp, _vp, _vp, _vp, _vp]),
    "kmsr_conv3x3_wino4_chan_sum_floats": (_sz, [C.POINTER(Conv3x3Desc)]),
    "kmsr_conv3x3_wino4_fwd_ex_f32": (C.c_int, [C.POINTER(Conv3x3Desc), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "kmsr_conv5x5_ps4_packed_weight_floats": (_sz, [_i32, _i32]),
    "kmsr_conv5x5_ps4_pack_weights_f32": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "kmsr_conv5x5_ps4_fwd_f32": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "kmsr_conv5x5_ps4_bf16_packed_weight_elems": (_sz, [_i32, _i32]),
    "kmsr_conv5x5_ps4_bf16_pack_weights": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "kmsr_conv5x5_ps4_fwd_bf16": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "kmsr_upsampler_frame_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "kmsr_conv_reflect_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "kmsr_convk_fwd_f32": (C.c_int, [C.POINTER(ConvKDesc), _vp, _vp, _vp, _vp, _vp]),
    "kmsr_convk_dgrad_f32": (C.c_int, [C.POINTER(ConvKDesc), _vp, _vp, _vp, _vp]),
    "kmsr_convk_dgrad_workspace_bytes": (C.c_size_t, [C.POINTER(ConvKDesc)]),
    "kmsr_convk_dgrad_ws_f32": (C.c_int, [C.POINTER(ConvKDesc), _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "kmsr_convk_wgrad_workspace_bytes": (_sz, [C.POINTER(ConvKDesc)]),
    "kmsr_convk_wgrad_f32": (C.c_int, [C.POINTER(ConvKDesc), _vp, _vp, _vp, _vp, _sz, _vp]),
    "kmsr_compose_kernels_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32),
                                           _vp, _vp]),
}


def load():
    """Load libkmsr_hip.so; raises KmsrError (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KmsrError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C {os.path.join(_HERE, 'csrc')}`).  There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


_EXC = {
    KMSR_E_BADARG: ValueError,
    KMSR_E_SHAPE: AssertionError,
    KMSR_E_HIP: RuntimeError,
    KMSR_E_RCCL: RuntimeError,
    KMSR_E_NOMEM: MemoryError,
    KMSR_E_UNSUPPORTED: NotImplementedError,
}


def check(rc):
    """Map a KMSR_E_* code back to the exception type the reference raises for that condition."""
    if rc != KMSR_OK:
        msg = load().kmsr_last_error().decode("utf-8", "replace")
        raise _EXC.get(rc, KmsrError)(f"kmsr rc={rc}: {msg}")


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise KmsrError("no MI355X visible (torch.cuda.is_available() is False); the HIP path has no CPU fallback")
    load()
    return torch.device("cuda", torch.cuda.current_device())


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
