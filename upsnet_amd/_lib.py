"""ctypes binding of libupsnet_hip.so (the C ABI declared in include/upsnet_hip.h).

The HIP library IS the product: there is no CPU / eager fallback. If the shared library is missing
(or a kernel launch fails) the ops raise RuntimeError.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UPSNET_LIB_PATH") or os.path.join(_HERE, "csrc", "libupsnet_hip.so")   # (the override: same-box A/B of two builds)

c_int, c_float, c_double, c_void_p, c_size_t, c_long = ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_long
P = c_void_p

_SIGNATURES = {
    "upsnet_last_error": (ctypes.c_char_p, []),
    "upsnet_last_kernel_form": (ctypes.c_char_p, []),
    "upsnet_abi_version": (c_int, []),
    "upsnet_zero_fill": (c_int, [P, P, c_size_t]),
    "upsnet_roi_align_forward": (c_int, [P, P, c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "upsnet_roi_align_forward_nhwc": (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, P, c_int, c_int, c_int, c_int, P]),
    "upsnet_fpn_roi_align_forward": (c_int, [P, P, P, P, P, c_int, P, c_int, P, c_int, c_int, c_int, P, P]),
    "upsnet_fpn_roi_order": (c_int, [P, P, c_int, P, c_int, c_int, P]),
    "upsnet_fpn_roi_align_forward_ordered": (c_int, [P, P, P, P, P, c_int, P, c_int, P, c_int, c_int, c_int, P, P, P]),
    "upsnet_roi_tuning": (None, [c_int]),
    "upsnet_roi_geometry": (None, [c_int, c_int]),
    "upsnet_deform_im2col": (c_int, [P, P, P] + [c_int] * 13 + [P]),
    "upsnet_mod_deform_im2col": (c_int, [P, P, P, P] + [c_int] * 15 + [P]),
    "upsnet_roi_align_backward": (c_int, [P, P, c_float] + [c_int] * 8 + [P, P]),
    "upsnet_deform_col2im": (c_int, [P, P, P] + [c_int] * 13 + [P]),
    "upsnet_deform_col2im_coord": (c_int, [P, P, P, P] + [c_int] * 13 + [P]),
    "upsnet_mod_deform_col2im": (c_int, [P, P, P, P] + [c_int] * 15 + [P]),
    "upsnet_mod_deform_col2im_coord": (c_int, [P, P, P, P, P] + [c_int] * 15 + [P, P]),
    "upsnet_deform_conv_forward_nhwc": (c_int, [P, c_int, P, P, P, P, P, P] + [c_int] * 11 + [P, c_int, P, c_int]),
    "upsnet_dcn_packed_weight_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "upsnet_dcn_pack_weight": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "upsnet_deform_conv_fused_nhwc": (c_int, [P, c_int, P, P, P, P, P, P] + [c_int] * 7 + [P, P, c_int]),
    "upsnet_deform_conv_fused_splitk_workspace_bytes": (c_size_t, [c_int] * 9),
    "upsnet_deform_conv_fused_nhwc_splitk": (c_int, [P, P, P, P, P] + [c_int] * 9 + [P, P, c_int, c_int, P]),
    "upsnet_dcn_tuning": (None, [c_int]),
    "upsnet_dcn_packed_weight_bf16_elems": (c_size_t, [c_int, c_int, c_int, c_int]),
    "upsnet_dcn_pack_weight_bf16": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "upsnet_deform_conv_fused_nhwc_bf16": (c_int, [P, c_int, P, P, P, P, P, P] + [c_int] * 7 + [P, P, c_int]),
    "upsnet_conv1x1_frag_nhwc_f32": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, P, c_int, c_int, c_int, c_int]),
    "upsnet_conv1x1_tuning": (None, [c_int]),
    "upsnet_conv1x1_siblings_nhwc_f32": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, P, c_int, c_int, c_int, c_int, c_int]),
    "upsnet_conv1x1_pair32_tuning": (None, [c_int]),
    "upsnet_conv1x1_splitk_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "upsnet_conv1x1_frag_nhwc_f32_splitk": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, P, c_int, c_int, c_int, c_int, P]),
    "upsnet_conv1x1_ksw_packed_weight_floats": (ctypes.c_size_t, [c_int, c_int]),
    "upsnet_conv1x1_ksw_pack_weight": (c_int, [P, P, c_int, c_int, P]),
    "upsnet_conv1x1_ksw_nhwc_f32": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int]),
    "upsnet_conv1x1_ksw_tuning": (None, [c_int]),
    "upsnet_conv3x3_ksw_packed_weight_floats": (ctypes.c_size_t, [c_int]),
    "upsnet_conv3x3_ksw_pack_weight": (c_int, [P, P, c_int, c_int, P]),
    "upsnet_conv3x3_ksw_nhwc_f32": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P, c_int, c_int]),
    "upsnet_conv2d_winograd36_splitk_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "upsnet_conv2d_winograd36_nhwc_f32_splitk": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, c_int, P, c_int, c_int, c_int, P]),
    "upsnet_conv1x1_pair_nhwc_f32": (c_int, [P, P, P, P, P, c_long, c_int, P, P, c_int, P, P, c_int]),
    "upsnet_conv2d_nhwc_f32": (c_int, [P, c_int, P, P, P, P, P, P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "upsnet_conv2d_nhwc_f32_multiw": (c_int, [P, c_int, P, P, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "upsnet_conv_tuning": (None, [c_int, c_int]),
    "upsnet_conv_pack_weight": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "upsnet_nms_host": (c_int, [P, P, P, c_int, c_int, c_float, c_int]),
    "upsnet_nms_workspace_bytes": (c_size_t, [c_int, c_int]),
    "upsnet_nms_batched": (c_int, [P, P, P, P, P, c_int, c_int, c_float, P, P, P]),
    "upsnet_nms_tuning": (None, [c_int]),
    "upsnet_cpu_nms_batched": (c_int, [P, P, P, P, c_int, c_int, c_double, P, P, P]),
    "upsnet_soft_nms_batched_workspace_bytes": (c_size_t, [c_int, c_int]),
    "upsnet_soft_nms_batched": (c_int, [P, P, P, P, c_int, c_int, c_float, c_float, c_float, c_int, P, P]),
    "upsnet_soft_nms_workspace_bytes": (c_size_t, [c_int]),
    "upsnet_soft_nms": (c_int, [P, P, P, c_int, c_float, c_float, c_float, c_int, P, P]),
    "upsnet_proposal_workspace_bytes": (c_size_t, [c_int, P, P, c_int, c_int, c_int]),
    "upsnet_pyramid_proposals_strided": (c_int, [P, c_int, P, P, P, P, P, P, P, P, P, P, c_int, P, c_int, c_int, c_float, c_float, P, P, P, P]),
    "upsnet_pyramid_proposals_strided_ordered": (c_int, [P, c_int, P, P, P, P, P, P, P, P, P, P, c_int, P, c_int, c_int, c_float, c_float, P, P, P, P, P]),
    "upsnet_pyramid_proposals_joint_strided": (c_int, [P, c_int, P, P, P, P, P, P, P, P, P, P, c_int, P, c_int, c_int, c_float, c_float, P, P, P, P]),
    "upsnet_pyramid_proposals": (c_int, [P, c_int, P, P, P, P, P, P, c_int, P, c_int, c_int, c_float, c_float, P, P, P, P]),
    "upsnet_mask_roi_capacity": (c_int, [c_int, c_int, c_int]),
    "upsnet_mask_roi_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "upsnet_mask_roi": (c_int, [P, P, P, P, c_int, P, c_int, P, c_int, c_float, c_float, c_int, P, P, P, P, P, P, P]),
    "upsnet_mask_roi_ex": (c_int, [P, P, P, P, c_int, P, c_int, P, c_int, c_int, c_float, c_float, c_int, P, P, P, P, P, P, P]),
    "upsnet_mask_logit_gather": (c_int, [P, P, c_int, c_int, c_int, c_long, c_long, c_long, P, P, c_int, P]),
    "upsnet_mask_removal_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "upsnet_mask_removal": (c_int, [P, P, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_double, P, P, P, P]),
    "upsnet_panoptic_tail_pack": (c_int, [P, P, P, c_int, P, P, P, P, P, P, P, P]),
    "upsnet_mask_paste": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "upsnet_seg_term": (c_int, [P, P, c_int, c_int, c_int, P, P, P, c_int, P]),
    "upsnet_panoptic_fuse": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, P, P, P, P, c_int, c_int, P, c_int, P, P]),
    "upsnet_panoptic_fuse_up": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, c_int, c_int, P, P, P]),
    "upsnet_conv2d_stem_nhwc4_f32": (c_int, [P, P, c_int, c_int, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "upsnet_conv_pack_weight_stem": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "upsnet_deconv2x2_nhwc_f32": (c_int, [P, P, c_int, c_int, c_int, c_int, P, c_int, P, c_int, c_int, P]),
    "upsnet_deconv2x2_pack_weight": (c_int, [P, P, c_int, c_int, c_int, P]),
    "upsnet_deconv2x2_frag_nhwc_f32": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P, c_int, c_int]),
    "upsnet_prep_image_u8": (c_int, [P, P, c_int, c_int, P, c_double, c_int, c_int, c_int, c_int, c_int, P]),
    "upsnet_image_to_nhwc4": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "upsnet_unified_pan_workspace_bytes": (c_size_t, []),
    "upsnet_unified_pan_result": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "upsnet_mask_roi_dedup": (c_int, [P, P, P, P, c_int, P, P, P, P, c_int, P, P, P]),
    "upsnet_conv2d_winograd_nhwc_f32": (c_int, [P, c_int, P, P, P, P, P, P, c_int, P, c_int, P, c_int, c_int]),
    "upsnet_conv2d_winograd_nhwc_f32_splitk": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, c_int, P, c_int, c_int, c_int, P]),
    "upsnet_conv_pack_weight_winograd": (c_int, [P, P, c_int, c_int, c_int, P]),
    "upsnet_conv_pack_weight_winograd_tn32": (c_int, [P, P, c_int, c_int, c_int, P]),
    "upsnet_conv2d_winograd_nhwc_f32_tn32": (c_int, [P, c_int, P, P, P, P, P, P, c_int, P, c_int, P, c_int, c_int]),
    "upsnet_conv2d_winograd_nhwc_f32_tail": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, c_int]),
    "upsnet_conv2d_winograd36_nhwc_f32": (c_int, [P, c_int, P, P, P, P, P, c_int, P, c_int, P, c_int, c_int]),
    "upsnet_conv_pack_weight_winograd36": (c_int, [P, P, c_int, c_int, c_int, P]),
    "upsnet_conv2d_nhwc_bf16": (c_int, [P, c_int, P, P, P, P, P, P, c_int, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int]),
    "upsnet_bottleneck_proj_bf16": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P]),
    "upsnet_conv_bf16_tuning": (c_int, [c_int, c_int]),
    "upsnet_conv1x1_bf16_tuning": (c_int, [c_int]),
    "upsnet_stem_pool_pack_weight_f32": (c_int, [P, P, c_int, P]),
    "upsnet_stem_pool_f32": (c_int, [P, P, c_int, c_int, c_int, P, P, P]),
    "upsnet_stem_pool_pack_weight_bf16": (c_int, [P, P, c_int, P]),
    "upsnet_stem_pool_bf16": (c_int, [P, P, c_int, c_int, c_int, P, P, P]),
    "upsnet_deconv2x2_nhwc_bf16": (c_int, [P, P, c_int, c_int, c_int, c_int, P, c_int, P, c_int, c_int, P, c_int]),
    "upsnet_bottleneck_bf16": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P, P, P, P, P]),
    "upsnet_conv_pack_weight_bf16": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "upsnet_conv2d_splitk_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "upsnet_conv2d_nhwc_f32_splitk": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "upsnet_im_post_rle": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "upsnet_fcn_score_combine": (c_int, [P, c_int, P, c_int, c_int, c_int, P, P]),
    "upsnet_panoptic_argmax": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, c_int, c_int, P]),
}

_lib = None


def exported_symbols():
    """Names declared in include/upsnet_hip.h (used by the CPU-side ABI test)."""
    return sorted(_SIGNATURES)


def lib():
    """Load libupsnet_hip.so; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("upsnet_amd: %s is missing -- build it with `python -m upsnet_amd.build` "
                               "(there is no CPU fallback)" % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("upsnet_amd.%s failed: %s" % (what, lib().upsnet_last_error().decode()))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise Exception("not implemented")  # same behaviour as functions/deform_conv.py:40-41


def f32c(t):
    """fp32 + contiguous (plumbing)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def nhwc(t):
    """Physical NHWC view of a logical NCHW tensor (plumbing; no copy if already channels_last)."""
    return t.contiguous(memory_format=torch.channels_last)


def int_array(vals):
    return (c_int * len(vals))(*[int(v) for v in vals])


def float_array(vals):
    return (c_float * len(vals))(*[float(v) for v in vals])


def ptr_array(tensors):
    return (c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])
