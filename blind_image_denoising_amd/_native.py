"""
ctypes binding of lib/libbfcnn_hip.so (the C ABI of include/bfcnn_hip.h).

There is deliberately NO fallback: if the HIP library is missing or fails to load, every
entry point of the package that needs it raises.  Build it with
`blind_image_denoising_amd/csrc/build.sh` (or `python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os
import pathlib

_HERE = pathlib.Path(__file__).resolve().parent
# BFCNN_HIP_LIB selects another build of the SAME library (tools/ablate.sh timing experiments)
LIB_PATH = pathlib.Path(os.environ.get("BFCNN_HIP_LIB", _HERE / "lib" / "libbfcnn_hip.so"))

BF_OK, BF_EINVAL, BF_EUNSUPPORTED, BF_EWORKSPACE, BF_EHIP = 0, -1, -2, -3, -4
BF_ACT_LINEAR, BF_ACT_RELU, BF_ACT_LEAKY_RELU = 0, 1, 2
BF_REG_NONE, BF_REG_L1, BF_REG_L2 = 0, 1, 2
BF_MODE_INFERENCE, BF_MODE_TRAIN = 0, 1
BF_LOSS_COUNT = 8
(BF_LOSS_TOTAL, BF_LOSS_DENOISER_TOTAL, BF_LOSS_MAE, BF_LOSS_MSE, BF_LOSS_SSIM,
 BF_LOSS_REGULARIZATION, BF_LOSS_MODEL_TOTAL, BF_LOSS_GRAD_NORM) = range(8)
EPI_RELU, EPI_AFFINE, EPI_RES, EPI_MASK, EPI_STATS, EPI_BNBWD = 1, 2, 4, 8, 16, 32


class ResnetDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "struct_size", "in_channels", "filters", "kernel_size", "no_layers", "block_convs",
        "block_kernel", "activation", "base_activation", "use_bn", "head_filters",
        "head_activation", "out_channels", "denormalize", "reg_base", "reg_block", "reg_head")] + \
        [(n, C.c_float) for n in ("v_min", "v_max", "bn_eps", "bn_momentum", "leaky_alpha")]


class LossDesc(C.Structure):
    _fields_ = [("struct_size", C.c_int32)] + [(n, C.c_float) for n in (
        "hinge", "cutoff", "mae_multiplier", "mse_multiplier", "ssim_multiplier",
        "regularization", "depth_weight")]


class TensorInfo(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("offset", C.c_int64), ("rank", C.c_int32),
                ("shape", C.c_int32 * 4), ("kind", C.c_int32), ("regularizer", C.c_int32)]


_P, _I, _I64, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); exactly the declarations of include/bfcnn_hip.h
SIGNATURES = {
    "bf_abi_version": (_I, []),
    "bf_create": (_I, [C.POINTER(ResnetDesc), C.POINTER(_P)]),
    "bf_destroy": (None, [_P]),
    "bf_last_error": (C.c_char_p, [_P]),
    "bf_param_count": (_I64, [_P]),
    "bf_state_count": (_I64, [_P]),
    "bf_tensor_count": (_I, [_P, _I]),
    "bf_tensor_at": (_I, [_P, _I, _I, C.POINTER(TensorInfo)]),
    "bf_packed_bytes": (_I64, [_P]),
    "bf_workspace_bytes": (_I64, [_P, _I, _I, _I, _I]),
    "bf_pack_inference": (_I, [_P, _P, _P, _P, _P]),
    "bf_forward_u8": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _I64, _P]),
    "bf_forward_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _I64, _P]),
    "bf_forward_u8_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _I64, _P]),
    "bf_train_step": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, C.POINTER(LossDesc), _P, _P, _P, _P, _I64, _P]),
    "bf_adam_step": (_I, [_P, _P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _F, _P, _P, _P]),
    "bf_adam_step_ex": (_I, [_P, _P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _F, _F, _P, _I, _P, _F, _P, _P, _P]),
    "bf_avgpool_s2_same": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "bf_avgpool2_valid": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "bf_upsample2x": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _P]),
    "bf_laplacian_split": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "bf_strided_slice2": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "bf_noise_augment": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _F, _F, C.c_uint64, _P]),
    "bf_op_pack_pointwise": (_I, [_P, _P, _I, _I, _P]),
    "bf_op_pointwise": (_I, [_P, _P, _P, _P, _P, C.c_int64, _I, _I, _I, _F, _P]),
    "bf_op_pointwise_ex": (_I, [_P, _P, _P, _P, _P, _P, C.c_int64, _I, _I, _I, _F, _I, _P]),
    "bf_op_convnext_mlp": (_I, [_P, _P, _P, _P, _P, _P, C.c_int64, _I, _I, _F, _P]),
    "bf_op_mlp_h3_pack_bytes": (C.c_int64, [_I]),
    "bf_op_pack_mlp_h3": (_I, [_P, _P, _P, _I, _P]),
    "bf_op_convnext_mlp_h3": (_I, [_P, _P, _P, _P, _P, C.c_int64, _I, _I, _F, _P]),
    "bf_op_convnext_block1_h3": (_I, [_P, _P, _P, _P, _F, _P, _P, C.c_int64, _I, _I, _F, _P]),
    "bf_op_pack_mlp_h3_chain": (_I, [_P, _P, _P, _I, _P]),
    "bf_op_convnext_chain32_h3": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _F, _I, _I, _I, _I, _F, _I, _F, _P]),
    "bf_op_convnext_block1_up_h3": (_I, [_P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _I, _F, _I, _F, _P]),
    "bf_op_convnext_block_h3": (_I, [_P, _P, _P, _I, _P, _F, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "bf_op_set_variant": (_I, [C.c_char_p, _I]),
    "bf_op_dwconv_ln": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _F, _P]),
    "bf_op_smooth_split": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "bf_op_conv2d": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    "bf_op_dwconv_mult": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    "bf_op_bneck_h3_pack_bytes": (C.c_int64, []),
    "bf_op_pack_bneck_h3": (_I, [_P, _P, _P, _P, _P]),
    "bf_op_bneck_block_h3": (_I, [_P, _P, _P, _P, _I, _F, _P, _I, _F, _P, _I, _F, _I, _I, _I, _I, _P]),
    "bf_op_dwmult_pointwise": (_I, [_P, _P, _P, _P, _I, _F, _P, _P, _I, _F, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "bf_op_maxpool2": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "bf_op_maxpool2_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "bf_op_norm_smooth_split": (_I, [_P, _P, _F, _I, _F, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "bf_op_upsample_act_add": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "bf_op_resize_bilinear": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "bf_op_attention": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "bf_op_attention_ld": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "bf_op_first_conv": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _I, _F, _P]),
    "bf_op_first_conv_h3": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _I, _F, _P]),
    "bf_op_first_conv_h3k": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _F, _I, _F, _P]),
    "bf_op_head_out": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P]),
    "bf_op_head_fused": (_I, [_P, _P, _F, _P, _I, _F, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P]),
    "bf_op_head_fused_h3": (_I, [_P, _P, _F, _P, _I, _F, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P]),
    "bf_op_fill32": (_I, [_P, _I, _I64, _P]),
    "bf_op_axpy": (_I, [_P, _P, _F, _I, _I64, _P]),
    "bf_op_conv2d_transpose": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    "bf_op_adam_step": (_I, [_P, _P, _P, _P, _I64, _I64, _F, _F, _F, _F, _F, _F, _F, _P, _I, _P, _F, _P, _P, _P]),
    "bf_op_act_bwd": (_I, [_P, _P, _P, _I64, _I, _F, _I, _P]),
    "bf_op_matmul_wgrad": (_I, [_P, _P, _P, _I64, _I, _I, _P, _I64, _P]),
    "bf_op_dwconv_wgrad": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _I64, _P]),
    "bf_op_layernorm_bwd": (_I, [_P, _P, _P, _P, _P, _I64, _I, _F, _P, _I64, _P]),
    "bf_op_scale_add": (_I, [_P, _P, _P, _P, _P, _I, _I64, _I, _P]),
    "bf_op_scale_add_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I64, _I, _P, _I64, _P]),
    "bf_op_multiplier_bwd": (_I, [_P, _P, _P, _I, _P]),
    "bf_op_smooth_split_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "bf_op_smooth_split_bwd_ex": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "bf_op_upsample2x_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "bf_op_conv2d_wgrad": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P, _I64, _P]),
    "bf_op_head_out_bwd": (_I, [_P, _P, _P, _P, _P, _I64, _I, _I, _I, _F, _F, _P, _I64, _P]),
    "bf_op_denoiser_loss_scratch_floats": (_I64, [_I, _I, _I, _I]),
    "bf_op_denoiser_loss": (_I, [_P, _P, _I, _I, _I, _I, C.POINTER(LossDesc), _P, _P, _P, _I64, _P]),
    "bf_op_attention_train": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "bf_op_attention_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "bf_op_resize_bilinear_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "bf_op_reg_elementwise": (_I, [_P, _P, _I64, _I, _F, _F, _P, _P]),
    "bf_op_reg_soft_orthonormal": (_I, [_P, _P, _I, _I, _F, _F, _F, _F, _P, _P, _P]),
    "bf_op_reg_soft_orthogonal_ex": (_I, [_P, _P, _I, _I, _F, _F, _F, _F, _P, _P, _I, _P]),
    "bf_op_bn_train_scratch_floats": (_I64, [_I]),
    "bf_op_bn_train_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I64, _I, _F, _F, _I, _F, _P, _I64, _P]),
    "bf_op_bn_train_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I64, _I, _P, _I64, _P]),
    "bf_op_gate_save_floats": (_I64, [_I, _I, _I]),
    "bf_op_gate_scratch_floats": (_I64, [_I, _I]),
    "bf_op_gate_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I64, _I, _I, _P, _I64, _P]),
    "bf_op_gate_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I64, _I, _I, _P, _I64, _P]),
    "bf_op_channel_gate_save_floats": (_I64, [_I, _I, _I, _I]),
    "bf_op_channel_gate_ex": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I64, _I, _I, _I, _I, _F, _I, _P, _I64, _P]),
    "bf_op_dense2": (_I, [_P, _P, _P, _P, _P, _P, _I64, _I, _I, _I, _I, _F, _I, _P]),
    "bf_op_selector_mix": (_I, [_P, _P, _P, _P, _I64, _I, _P]),
    "bf_op_avgpool_same": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "bf_op_concat_channels": (_I, [_P, _P, _P, _P, _I64, _I, _I, _I, _P]),
    "bf_op_channel_mean_broadcast": (_I, [_P, _P, _I, _I64, _I, _I64, _P, _I64, _P]),
    "bf_op_sigmoid_gate": (_I, [_P, _P, _P, _P, _I64, _P]),
    "bf_op_sigmoid_gate_bwd": (_I, [_P, _P, _P, _P, _P, _I64, _P]),
    "bf_op_relu_shift": (_I, [_P, _I, _F, _P, _I, _P]),
    "bf_op_linear_shift": (_I, [_P, _I, _F, _P, _I, _P]),
    "bf_op_center_scale": (_I, [_P, _P, _P, _P, _I64, _F, _P]),
    "bf_op_pass_filter": (_I, [_P, _P, _I64, _F, _I, _I, _P]),
    "bf_op_pass_filter_bwd": (_I, [_P, _P, _P, _I64, _F, _I, _I, _P]),
    "bf_op_center_scale_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I64, _F, _P]),
    "bf_op_center_sq_bwd": (_I, [_P, _P, _P, _P, _P, _I64, _P]),
    "bf_op_concat_input": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P]),
    "bf_op_selector_mix_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I64, _I, _P]),
    "bf_op_avgpool_same_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "bf_op_dense2_bwd_scratch_floats": (_I64, [_I64, _I, _I]),
    "bf_op_dense2_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I64, _I, _I, _I, _F, _P, _I64, _P]),
    "bf_op_slice_channels": (_I, [_P, _P, _I64, _I, _I, _I, _P]),
    "bf_op_relu_shift_bwd": (_I, [_P, _I, _F, _P, _P, _I, _P]),
    "bf_op_normalize": (_I, [_P, _P, _I64, _F, _F, _I, _P]),
    "bf_op_channel_repeat": (_I, [_P, _P, _I64, _I, _I, _P]),
    "bf_op_channel_group_sum": (_I, [_P, _P, _I64, _I, _I, _P]),
    "bf_op_group_kernel": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "bf_op_flip_hw": (_I, [_P, _P, _I, _I, _P]),
    "bf_op_transpose2d": (_I, [_P, _P, _I, _I, _P]),
    "bf_comm_unique_id": (_I, [C.c_char_p]),
    "bf_comm_init_rank": (_I, [C.POINTER(_P), _I, _I, C.c_char_p]),
    "bf_comm_destroy": (_I, [_P]),
    "bf_allreduce_grads": (_I, [_P, _P, _I64, _P, _P]),
    "bf_comm_last_error": (C.c_char_p, []),
    "bf_op_channel_multiplier": (_I, [_P, _P, _I, _P]),
    "bf_set_option": (_I, [_P, C.c_char_p, _I]),
    "bf_get_timing": (_I, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "bf_get_block_kernel": (C.c_char_p, [_P, C.POINTER(C.c_int)]),
    "bf_get_train_kernels": (C.c_char_p, [_P]),
    "bf_debug_conv3x3": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "bf_debug_conv3x3_grid": (_I, [_I, _I, _I]),
    "bf_debug_fused_block": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "bf_debug_fused_block_h3_scratch_floats": (_I64, [_I, _I, _I]),
    "bf_debug_fused_block_h3": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "bf_debug_fused_block2_h3_scratch_floats": (_I64, [_I, _I, _I]),
    "bf_debug_fused_block2_h3": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "bf_debug_set_h3_variant": (_I, [_I]),
    "bf_debug_set_upsample_band": (_I, [_I]),
    "bf_debug_conv3x3_h3_scratch_floats": (_I64, []),
    "bf_debug_conv3x3_h3": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "bf_debug_wgrad_partial_floats": (_I64, [_I, _I, _I]),
    "bf_debug_wgrad3x3": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "bf_debug_wgrad3x3_h3": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "bf_debug_fwd_block_h3t_scratch_floats": (_I64, [_I, _I, _I]),
    "bf_debug_fwd_block_h3t": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "bf_debug_bwd_block_h3t_scratch_floats": (_I64, [_I, _I, _I]),
    "bf_debug_bwd_block_h3t": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "bf_debug_conv3x3_h3_pre": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "bf_debug_bwd3x3_h3_scratch_floats": (_I64, [_I, _I, _I]),
    "bf_debug_bwd3x3_h3_grid": (_I, [_I, _I, _I]),
    "bf_debug_bwd3x3_h3_grid_ex": (_I, [_I, _I, _I, _I]),
    "bf_debug_bwd3x3_h3": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "bf_debug_mfma_probe": (_I, [_P, _P, _P, _P]),
}

BF_STATUS_BYTES = 2048         # tail of an inference workspace: int32 status word + kernel scratch
BF_STATUS_F16_RANGE = 1

_lib = None


class StreamArg(C.c_void_p):
    """hipStream_t argument that remembers which device it belongs to (see _device_guard)."""
    _bf_device = None


def _device_guard(fn):
    """A HIP launch goes to the CURRENT device of the calling thread, whatever memory its pointers name; a stream of
    another device is an error, and the null stream silently means "the current device".  Every launching entry point
    takes the stream `stream_ptr(tensor)` made from one of its tensors: the wrapper runs the call with that tensor's
    device current (a no-op in the usual one-process-per-GPU setting)."""
    def call(*args):
        for a in args:
            dev = getattr(a, "_bf_device", None)
            if dev is not None:
                import torch
                if torch.cuda.current_device() != dev:
                    with torch.cuda.device(dev):
                        return fn(*args)
                break
        return fn(*args)
    call.raw = fn
    call.__name__ = getattr(fn, "__name__", "bf_call")
    return call


class _Library:
    """namespace of the guarded entry points (attribute access as on the ctypes.CDLL it wraps)."""

    def __init__(self, handle):
        self._handle = handle

    def __getattr__(self, name):          # symbols outside SIGNATURES (tools/: debug hooks): the raw ctypes function
        return getattr(self._handle, name)


def lib():
    """The loaded library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} is missing: the gfx950 HIP library has not been built "
                f"(run blind_image_denoising_amd/csrc/build.sh). There is no CPU fallback.")
        handle = C.CDLL(str(LIB_PATH), mode=getattr(os, "RTLD_NOW", 2))
        wrapped = _Library(handle)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
            setattr(wrapped, name, _device_guard(fn))
        if handle.bf_abi_version() != 1:
            raise ImportError("libbfcnn_hip.so ABI version mismatch")
        _lib = wrapped
    return _lib


def last_error(handle=None) -> str:
    msg = lib().bf_last_error(handle)
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, handle=None, what: str = ""):
    """Maps the C status to the exception the reference raises for the same condition."""
    if rc == BF_OK:
        return
    msg = f"{what}: {last_error(handle)}" if what else last_error(handle)
    if rc == BF_EINVAL:
        raise ValueError(msg)
    if rc == BF_EUNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == BF_EWORKSPACE:
        raise MemoryError(msg)
    raise RuntimeError(f"HIP error ({rc}) {msg}")


def ptr(t):
    """Device (or host) address of a torch tensor / None."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def stream_ptr(t=None):
    """hipStream_t of torch's current stream on the tensor's device (and, for the call it is passed to, the device
    that call runs on: _device_guard)."""
    import torch
    if t is not None and t.is_cuda:
        dev = t.device.index if t.device.index is not None else torch.cuda.current_device()
    else:
        dev = torch.cuda.current_device()
    s = StreamArg(torch.cuda.current_stream(dev).cuda_stream)
    s._bf_device = dev
    return s
