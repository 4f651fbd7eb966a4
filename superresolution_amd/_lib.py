"""ctypes binding of libsrhip.so (declared in include/srhip.h)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SRHIP_LIBRARY selects another build of the same ABI (kernel experiments); default: the in-tree library
LIB_PATH = os.environ.get("SRHIP_LIBRARY") or os.path.join(_HERE, "libsrhip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C superresolution_amd/csrc` "
        "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")


def _load_hip_runtime() -> None:
    """libsrhip.so carries no DT_NEEDED for the HIP runtime: it must bind to the ONE runtime this
    process uses for device memory and streams, which is the copy PyTorch-ROCm loaded.  Promote
    that copy to the global symbol scope before loading the kernels."""
    import torch  # noqa: F401  (loads torch/lib/libamdhip64.so)
    cand = [os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"),
            "/opt/rocm/lib/libamdhip64.so"]
    for path in cand:
        if os.path.exists(path):
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            return
    raise ImportError("no libamdhip64.so found (looked in torch/lib and /opt/rocm/lib)")


_load_hip_runtime()
lib = ctypes.CDLL(LIB_PATH)

c_int, c_float, c_double, c_void_p, c_size_t = (ctypes.c_int, ctypes.c_float, ctypes.c_double,
                                                ctypes.c_void_p, ctypes.c_size_t)


class Conv3x3Args(ctypes.Structure):
    """srhip_conv3x3_args"""
    _fields_ = [
        ("src_a", c_void_p), ("src_b", c_void_p), ("wpack", c_void_p), ("bias", c_void_p),
        ("res", c_void_p), ("dst", c_void_p),
        ("n", c_int), ("h", c_int), ("w", c_int),
        ("ca", c_int), ("cb", c_int),
        ("cout", c_int), ("cout_stride", c_int),
        ("act", c_int), ("slope", c_float), ("alpha", c_float),
        ("dst_nchw_f32", c_int), ("res_mode", c_int), ("ca_stride", c_int), ("cout_w", c_int),
        ("aux", c_void_p), ("aux_mode", c_int), ("dst2", c_void_p),
    ]


class Conv3x3PairArgs(ctypes.Structure):
    """srhip_conv3x3_pair_args"""
    _fields_ = [
        ("src", c_void_p), ("wpack1", c_void_p), ("bias1", c_void_p), ("wpack2", c_void_p), ("bias2", c_void_p),
        ("mid", c_void_p), ("dst", c_void_p),
        ("n", c_int), ("h", c_int), ("w", c_int),
        ("act2", c_int), ("slope2", c_float), ("alpha", c_float), ("cout2", c_int),
        ("mid_bits", c_void_p), ("mid_bits_mode", c_int),
    ]


class ConvGemmArgs(ctypes.Structure):
    """srhip_conv_gemm_args"""
    _fields_ = [
        ("src", c_void_p * 3), ("c", c_int * 3), ("cstride", c_int * 3), ("nsrc", c_int),
        ("wpack", c_void_p), ("bias", c_void_p), ("res", c_void_p), ("dst", c_void_p),
        ("n", c_int), ("h", c_int), ("w", c_int), ("cout", c_int), ("cout_stride", c_int),
        ("ksize", c_int), ("stride", c_int), ("pad", c_int), ("act", c_int), ("slope", c_float), ("alpha", c_float),
        ("res_mode", c_int),
    ]


class WgradJob(ctypes.Structure):
    """srhip_wgrad_job"""
    _fields_ = [
        ("x", c_void_p), ("dz", c_void_p), ("dw", c_void_p), ("db", c_void_p),
        ("ci_real", c_int), ("ci_off", c_int), ("co_real", c_int),
        ("x_cstride", c_int), ("x_coff", c_int), ("dz_cstride", c_int), ("dz_coff", c_int), ("reserved", c_int),
    ]


class Conv1x1ChainArgs(ctypes.Structure):
    """srhip_conv1x1_chain_args"""
    _fields_ = [
        ("src", c_void_p), ("res", c_void_p), ("dst", c_void_p), ("wpack", c_void_p * 3), ("bias", c_void_p * 3),
        ("npix", ctypes.c_longlong), ("c", c_int * 4), ("nlayers", c_int), ("act", c_int * 3), ("slope", c_float), ("alpha", c_float),
    ]


_SIGNATURES = {
    "srhip_version": (c_int, []),
    "srhip_last_error": (ctypes.c_char_p, []),
    "srhip_nchw_f32_to_nhwc_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "srhip_nhwc_bf16_to_nchw_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "srhip_conv3x3_packed_bytes": (c_size_t, [c_int, c_int]),
    "srhip_conv3x3_signbits_bytes": (c_size_t, [c_int, c_int, c_int]),
    "srhip_conv3x3_pack_weights": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                           c_int, c_int, c_void_p]),
    "srhip_conv3x3_fwd": (c_int, [ctypes.POINTER(Conv3x3Args), c_void_p]),
    "srhip_conv3x3_pair_fwd": (c_int, [ctypes.POINTER(Conv3x3PairArgs), c_void_p]),
    "srhip_conv3x3_pair_bits_bytes": (c_size_t, [c_int, c_int, c_int]),
    "srhip_conv_gemm_packed_bytes": (c_size_t, [c_int, c_int, c_int]),
    "srhip_conv_gemm_pack_weights": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "srhip_conv_gemm_pack_weights_transposed": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "srhip_conv_gemm_fwd": (c_int, [ctypes.POINTER(ConvGemmArgs), c_void_p]),
    "srhip_conv1x1_chain_fwd": (c_int, [ctypes.POINTER(Conv1x1ChainArgs), c_void_p]),
    "srhip_conv1x1_chain_supported": (c_int, [ctypes.POINTER(c_int), c_int, c_int]),
    "srhip_conv3x3_wgrad_workspace_bytes": (c_size_t, []),
    "srhip_conv3x3_wgrad": (c_int, [c_void_p] * 5 + [c_int] * 10 + [c_float, c_void_p]),
    "srhip_conv3x3_wgrad_windows": (c_int, [c_void_p] * 5 + [c_int] * 10 + [c_float] + [c_int] * 4 + [c_void_p]),
    "srhip_conv3x3_wgrad_multi": (c_int, [ctypes.POINTER(WgradJob), c_int, c_void_p] + [c_int] * 7 + [c_float, c_void_p]),
    "srhip_conv1x1_wgrad_multi": (c_int, [ctypes.POINTER(WgradJob), c_int, c_void_p] + [c_int] * 7 + [c_float, c_void_p]),
    "srhip_conv2d_direct": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 13 +
                            [c_int, c_float, c_float, c_void_p]),
    "srhip_upsample_grid_bilinear_nhwc": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "srhip_upsample_grid_bilinear_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "srhip_upsample_grid_bilinear_bwd_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "srhip_upsample_grid_bilinear_bwd_nhwc": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "srhip_relu_mask_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "srhip_signbits_from_map_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "srhip_add_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "srhip_scale_add_bf16": (c_int, [c_void_p] * 4 + [c_size_t, c_void_p]),
    "srhip_scale_add_bwd_bf16": (c_int, [c_void_p] * 5 + [c_int, c_size_t, c_void_p]),
    "srhip_max_pool2d_nhwc": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "srhip_resize_bilinear_nhwc": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "srhip_conv_transpose2x2s2": (c_int, [c_void_p] * 4 + [c_int] * 8 + [c_void_p]),
    "srhip_gradient_map_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "srhip_sigmoid_gate_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "srhip_esa_tail_bf16": (c_int, [c_void_p] * 8 + [c_int] * 7 + [c_void_p]),
    "srhip_l1_ssim_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "srhip_l1_ssim_loss": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_float] * 4 + [c_void_p]),
    "srhip_weighted_l1_workspace_bytes": (c_size_t, []),
    "srhip_weighted_l1_loss": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_void_p]),
    "srhip_backward_warp_f32": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "srhip_warp_concat_nchw_f32_to_nhwc_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "srhip_backward_warp_nhwc": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "srhip_backward_warp_bwd_f32": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "srhip_pixel_shuffle": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "srhip_pixel_unshuffle4_nchw_f32_to_nhwc_bf16": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "srhip_gradient_map_unshuffle4_nhwc_bf16": (c_int, [c_void_p, c_int, ctypes.POINTER(c_int), c_int, c_void_p] + [c_int] * 3 + [c_float, c_void_p]),
    "srhip_pixel_shuffle4_conv3x3_c3_nchw_f32": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_float, c_void_p]),
    "srhip_pixel_shuffle4_nhwc_bf16_to_nhwc16": (c_int, [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "srhip_pixel_shuffle4_nhwc16_bwd_bf16": (c_int, [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "srhip_zero_insert2_nhwc_bf16": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "srhip_pixel_unshuffle4_nchw_f32_to_nhwc_bf16_max": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p, c_int, c_void_p]),
    "srhip_pixel_unshuffle": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "srhip_pixel_shuffle_nhwc_to_nchw_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "srhip_nearest_down": (c_int, [c_void_p, c_void_p] + [c_int] * 5 + [c_double, c_int, c_void_p]),
    "srhip_rel_coord": (c_int, [c_void_p] + [c_int] * 4 + [c_void_p]),
    "srhip_make_coord": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "srhip_concat_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "srhip_layernorm_cf_f32": (c_int, [c_void_p] * 4 + [c_int] * 3 + [c_float, c_void_p]),
    "srhip_depth_to_space2_nhwc": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "srhip_lrelu_mask_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_void_p]),
    "srhip_sigmoid_gate_bwd_bf16": (c_int, [c_void_p] * 5 + [c_size_t, c_void_p]),
    "srhip_max_pool2d_bwd_nhwc": (c_int, [c_void_p] * 3 + [c_int] * 6 + [c_void_p]),
    "srhip_dilate_velocity_f32": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "srhip_attention_fuse_bf16": (c_int, [c_void_p] * 7 + [c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "srhip_attention_fuse_bwd_bf16": (c_int, [c_void_p] * 7 + [c_int] + [c_void_p] * 6 + [c_int, c_size_t, c_int, c_void_p]),
    "srhip_max_pool2d_arg_nhwc": (c_int, [c_void_p] * 3 + [c_int] * 6 + [c_void_p]),
    "srhip_max_pool2d_bwd_arg_nhwc": (c_int, [c_void_p] * 3 + [c_int] * 6 + [c_void_p]),
    "srhip_resize_bilinear_bwd_nhwc": (c_int, [c_void_p] * 2 + [c_int] * 7 + [c_void_p]),
    "srhip_crop_flip_f16_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 9 + [ctypes.c_uint, c_void_p]),
    "srhip_disocclusion_f32": (c_int, [c_void_p] * 8 + [c_int] * 4 + [c_void_p]),
    "srhip_clamp_depth_f32": (c_int, [c_void_p, c_void_p, ctypes.c_longlong, c_float, c_float, c_void_p]),
    "srhip_layernorm_cf_bwd_workspace_bytes": (c_size_t, [c_int] * 3),
    "srhip_layernorm_cf_bwd_f32": (c_int, [c_void_p] * 7 + [c_int] * 3 + [c_float, c_void_p]),
    "srhip_layernorm_cf_bwd_f32_static": (c_int, [c_void_p] * 6 + [c_int] * 3 + [c_float, c_void_p]),
}

for _name, (_res, _args) in _SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTED = tuple(_SIGNATURES)


class SrhipError(RuntimeError):
    pass


def check(status: int) -> None:
    if status != 0:
        raise SrhipError(f"libsrhip status {status}: {lib.srhip_last_error().decode()}")
