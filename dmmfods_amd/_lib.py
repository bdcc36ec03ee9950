"""ctypes binding of libdmmfods_hip.so.  There is no CPU fallback: if the library is missing or a call
fails, this raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DMM_LIB_PATH") or os.path.join(_HERE, "libdmmfods_hip.so")  # DMM_LIB_PATH: experiment builds

DMM_F32, DMM_F16, DMM_BF16 = 0, 1, 2
LOSS_BCE, LOSS_FOCAL = 0, 1
T_CONV, T_CONVT, T_BN_WEIGHT, T_BN_BIAS, T_BN_MEAN, T_BN_VAR, T_BN_TRACKED = range(7)
ERR_INVALID, ERR_SHAPE, ERR_HIP, ERR_STATE, ERR_NO_DEVICE = -1, -2, -3, -4, -5


class ModelDesc(C.Structure):
    _fields_ = [
        ("growth_rate", C.c_int32), ("num_blocks", C.c_int32), ("block_config", C.c_int32 * 8),
        ("num_init_features", C.c_int32), ("bn_size", C.c_int32), ("num_classes", C.c_int32),
        ("concat_before_block_num", C.c_int32), ("stream_1_in_channels", C.c_int32),
        ("stream_2_in_channels", C.c_int32), ("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
        ("dtype", C.c_int32), ("loss_scale", C.c_float), ("bn_momentum", C.c_float), ("bn_eps", C.c_float),
        ("iou_threshold", C.c_float), ("use_mfma", C.c_int32),
    ]


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dtype", "use_mfma", "B", "H", "W", "Cin", "Cout", "R", "S", "stride", "pad", "transposed", "mode", "bn_relu")]


class DmmError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C dmmfods_amd/csrc`).  dmmfods_amd has no CPU fallback.")
    # PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so).  Device pointers and streams are only
    # meaningful inside ONE runtime, so make sure torch's copy is the one already loaded when our library's
    # NEEDED libamdhip64.so.7 is resolved (same SONAME -> the loader reuses it).
    import torch  # noqa: F401
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tl):
        C.CDLL(tl, mode=C.RTLD_GLOBAL)
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
    L.dmm_last_error.restype = C.c_char_p
    L.dmm_version.restype = C.c_int
    L.dmm_last_impl.restype = C.c_int
    L.dmm_impl_name.restype = C.c_char_p
    L.dmm_impl_name.argtypes = [C.c_int]
    L.dmm_impl_mask.restype = C.c_uint
    L.dmm_impl_mask.argtypes = [C.c_int]
    L.dmm_set_option.restype = C.c_int
    L.dmm_set_option.argtypes = [C.c_char_p, C.c_int]
    L.dmm_plan_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(vp)]
    L.dmm_plan_destroy.argtypes = [vp]
    L.dmm_plan_destroy.restype = C.c_int
    L.dmm_plan_num_tensors.argtypes = [vp]
    L.dmm_plan_tensor_info.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(i32), C.POINTER(i32),
                                       C.POINTER(i64 * 4), C.POINTER(i64)]
    L.dmm_plan_num_params.argtypes = [vp]
    L.dmm_plan_num_params.restype = i64
    L.dmm_plan_num_buffer_elems.argtypes = [vp]
    L.dmm_plan_num_buffer_elems.restype = i64
    L.dmm_plan_workspace_bytes.argtypes = [vp]
    L.dmm_plan_workspace_bytes.restype = sz
    L.dmm_plan_forward_flops.argtypes = [vp]
    L.dmm_plan_forward_flops.restype = C.c_double
    L.dmm_plan_bind.argtypes = [vp, vp, sz, vp, vp, vp]
    L.dmm_plan_forward.argtypes = [vp, vp, vp, vp, C.c_int, vp]
    L.dmm_plan_loss_backward.argtypes = [vp, vp, vp, vp, vp]
    L.dmm_plan_backward.argtypes = [vp, vp, vp]
    L.dmm_plan_loss_metrics.argtypes = [vp, vp, vp, vp, vp]
    L.dmm_plan_set_loss.argtypes = [vp, C.c_int, vp, vp, C.c_int]
    L.dmm_loss_forward.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.dmm_plan_num_grad_buckets.argtypes = [vp]
    L.dmm_plan_num_graph_replays.argtypes = [vp, C.c_int]
    L.dmm_plan_num_graph_replays.restype = C.c_longlong
    L.dmm_plan_grad_bucket.argtypes = [vp, C.c_int, C.POINTER(i64), C.POINTER(i64)]
    L.dmm_plan_grad_bucket_wait.argtypes = [vp, C.c_int, vp]
    L.dmm_plan_profile_begin.argtypes = [vp, C.c_int]
    L.dmm_plan_profile_filter.restype = C.c_int
    L.dmm_plan_profile_filter.argtypes = [vp, C.c_char_p]
    L.dmm_plan_profile_num_ops.argtypes = [vp, C.c_int]
    L.dmm_plan_profile_op.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.dmm_plan_profile_collect.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
    L.dmm_adam_step.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i64, f32, vp]
    L.dmm_conv_scratch_bytes.argtypes = [C.POINTER(ConvDesc)]
    L.dmm_conv_scratch_bytes.restype = sz
    L.dmm_conv_forward.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp]
    L.dmm_conv_wgrad.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp]
    L.dmm_conv_dgrad.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.dmm_conv_wgrad_ex.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp]
    L.dmm_conv_dgrad_ex.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp]
    L.dmm_conv1x1_backward_fused.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp]
    L.dmm_conv5_wgrad_stats.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp]
    _lib = L
    return L


EXPORTS = [
    "dmm_last_error", "dmm_version", "dmm_set_option", "dmm_plan_create", "dmm_plan_destroy", "dmm_plan_num_tensors",
    "dmm_plan_tensor_info", "dmm_plan_num_params", "dmm_plan_num_buffer_elems", "dmm_plan_workspace_bytes",
    "dmm_plan_forward_flops", "dmm_plan_bind", "dmm_plan_forward", "dmm_plan_loss_backward", "dmm_plan_backward",
    "dmm_plan_loss_metrics", "dmm_plan_num_graph_replays", "dmm_plan_set_loss", "dmm_loss_forward", "dmm_plan_num_grad_buckets", "dmm_plan_grad_bucket",
    "dmm_plan_grad_bucket_wait", "dmm_plan_profile_begin", "dmm_plan_profile_filter", "dmm_plan_profile_num_ops", "dmm_plan_profile_op",
    "dmm_plan_profile_collect", "dmm_adam_step", "dmm_conv_scratch_bytes", "dmm_conv_forward", "dmm_conv_wgrad",
    "dmm_conv_dgrad", "dmm_conv_wgrad_ex", "dmm_conv_dgrad_ex", "dmm_conv1x1_backward_fused", "dmm_conv5_wgrad_stats", "dmm_last_impl", "dmm_impl_name", "dmm_impl_mask",
]


def check(rc):
    """Map a dmm_status to the exception type the reference raises for the same condition."""
    if rc == 0:
        return
    msg = lib().dmm_last_error().decode(errors="replace")
    if rc == ERR_SHAPE:
        raise ValueError(msg)          # reference: ValueError from ConvTranspose2d(output_size=...) (M:261)
    if rc == ERR_INVALID and "fusion" in msg:
        raise AttributeError(msg)      # reference: AttributeError for a bad fusion configuration (M:65)
    if rc == ERR_INVALID:
        raise ValueError(msg)
    raise DmmError(f"dmm status {rc}: {msg}")


def last_impl():
    """Name of the kernel family that ran this thread's most recent single-kernel launch ("wg3", "conv3", "generic", ...)."""
    L = lib()
    return L.dmm_impl_name(L.dmm_last_impl()).decode()


def impls_since_reset(reset=True):
    """Names of the kernel families that ran this thread's single-kernel launches since the last reset."""
    L = lib()
    m = L.dmm_impl_mask(1 if reset else 0)
    return {L.dmm_impl_name(i).decode() for i in range(32) if (m >> i) & 1}


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
