"""ctypes binding of libp3d_hip.so (the C ABI declared in include/p3d_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, the
caller gets a P3DError.  PyTorch is used only to own device memory and streams.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('P3D_LIB') or os.path.join(_HERE, 'csrc', 'libp3d_hip.so')   # P3D_LIB: A/B a tuning build
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'p3d_hip.h')


class P3DError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    """struct p3d_conv_desc"""
    _fields_ = [(n, ctypes.c_int32) for n in
                ('N', 'C', 'H', 'W', 'K', 'R', 'S', 'stride', 'pad', 'dil', 'Ho', 'Wo', 'c_total', 'c_offset',
                 'accumulate', 'reserved')]


_i32, _i64, _f32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float
_ptr, _sz = ctypes.c_void_p, ctypes.c_size_t
_desc = ctypes.POINTER(ConvDesc)

# name -> (restype, argtypes); must list every function of include/p3d_hip.h (tests/test_abi.py checks)
SIGNATURES = {
    'p3d_version': (_i32, []),
    'p3d_last_error': (ctypes.c_char_p, []),
    'p3d_conv2d_fwd': (_i32, [_desc, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _sz, _ptr]),
    'p3d_conv2d_fwd_workspace_bytes': (_sz, [_desc]),
    'p3d_conv2d_bn_eval_fwd_workspace_bytes': (_sz, [_desc]),
    'p3d_conv2d_bn_eval_fwd': (_i32, [_desc, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _f32, _ptr, _i32, _ptr, _ptr, _sz, _ptr]),
    'p3d_conv2d_dgrad_workspace_bytes': (_sz, [_desc]),
    'p3d_conv2d_dgrad': (_i32, [_desc, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _sz, _ptr]),
    'p3d_conv2d_wgrad_workspace_bytes': (_sz, [_desc]),
    'p3d_conv2d_wgrad': (_i32, [_desc, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _sz, _ptr]),
    'p3d_conv2d_bgrad': (_i32, [_ptr, _i32, _i32, _i32, _ptr, _i32, _ptr]),
    'p3d_conv2d_bgrad_masked': (_i32, [_ptr, _ptr, _i32, _i32, _i32, _ptr, _i32, _ptr]),
    'p3d_block_supported': (_i32, [_ptr]),
    'p3d_block_tail_supported': (_i32, [_ptr]),
    'p3d_block_tail_partial_bytes': (ctypes.c_size_t, [_ptr]),
    'p3d_block_workspace_bytes': (_i32, [_ptr, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    'p3d_block_fwd': (_i32, [_ptr, _ptr, _ptr, _sz, _ptr]),
    'p3d_block_bwd': (_i32, [_ptr, _ptr, _ptr, _sz, _ptr, _sz, _ptr, _ptr]),
    'p3d_fx_weight_image_bytes': (_i32, [_i32, _i32, _i32, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    'p3d_fx_weight_images': (_i32, [_ptr, _i32, _i32, _i32, _ptr, _ptr, _ptr]),
    'p3d_fx_weight_images_batched': (_i32, [_ptr, _i32, _i32, _ptr]),
    'p3d_hblock_workspace_bytes': (_i32, [_ptr, _ptr, _ptr]),
    'p3d_hblock_fwd': (_i32, [_ptr, _ptr, _ptr, _sz, _ptr]),
    'p3d_hblock_bwd': (_i32, [_ptr, _ptr, _ptr, _sz, _ptr, _sz, _ptr, _ptr]),
    'p3d_fx_act_image_bytes': (_sz, [_i32, _i32, _i32]),
    'p3d_fx_act_image': (_i32, [_i32, _ptr, _ptr, _ptr, _i32, _ptr, _i32, _i32, _i32, _ptr]),
    'p3d_fx_conv_img_workspace_bytes': (_sz, [_ptr, _i32]),
    'p3d_fx_conv_img_supported': (_i32, [_ptr]),
    'p3d_fx_conv_fwd_img': (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _sz, _ptr]),
    'p3d_fx_conv_dgrad_img': (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _sz, _ptr]),
    'p3d_fx_conv_wgrad_img': (_i32, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _sz, _ptr]),
    'p3d_stem_supported': (_i32, [_i32] * 5),
    'p3d_stem_image_bytes': (_sz, [_i32] * 3),
    'p3d_stem_weight_image_bytes': (_sz, [_i32]),
    'p3d_stem_workspace_bytes': (_sz, [_i32] * 4),
    'p3d_stem_image': (_i32, [_ptr, _ptr, _i32, _i32, _i32, _i32, _ptr]),
    'p3d_stem_weight_image': (_i32, [_ptr, _i32, _i32, _ptr, _ptr, _sz, _ptr]),
    'p3d_stem_fwd': (_i32, [_ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _ptr]),
    'p3d_stem_wgrad': (_i32, [_ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _i32, _ptr, _sz, _ptr]),
    'p3d_stem_masked_supported': (_i32, [_i32] * 5),
    'p3d_stem_image_masked': (_i32, [_ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _ptr]),
    'p3d_stem_fwd_masked': (_i32, [_ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _ptr]),
    'p3d_stem_wgrad_masked': (_i32, [_ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _i32, _ptr, _sz, _ptr]),
    'p3d_profile_enable': (_i32, [_i32]),
    'p3d_profile_collect': (_i32, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]),
    'p3d_profile_collect2': (_i32, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]),
    'p3d_mask_count_fwd': (_i32, [_desc, _ptr, _ptr, _ptr, _ptr]),
    'p3d_nonzero_mask': (_i32, [_ptr, _ptr, _i64, _ptr]),
    'p3d_bn_workspace_bytes': (_sz, [_i32, _i32, _i32]),
    'p3d_bn_train_fwd': (_i32, [_ptr] * 9 + [_i32, _i32, _i32, _f32, _f32, _i32, _ptr, _sz, _ptr]),
    'p3d_bn_train_bwd': (_i32, [_ptr] * 11 + [_i32, _i32, _i32, _i32, _i32, _ptr, _sz, _ptr]),
    'p3d_bn_eval_fwd': (_i32, [_ptr] * 7 + [_i32, _i32, _i32, _f32, _i32, _ptr]),
    'p3d_bn_eval_bwd': (_i32, [_ptr] * 10 + [_i32, _i32, _i32, _f32, _i32, _i32, _ptr, _sz, _ptr]),
    'p3d_relu_fwd': (_i32, [_ptr, _ptr, _i64, _ptr]),
    'p3d_relu_bwd': (_i32, [_ptr, _ptr, _ptr, _i64, _ptr]),
    'p3d_maxpool3x3s2_fwd': (_i32, [_ptr, _ptr, _ptr, _i32, _i32, _i32, _ptr]),
    'p3d_maxpool3x3s2_bwd': (_i32, [_ptr, _ptr, _ptr, _i32, _i32, _i32, _ptr]),
    'p3d_stem_tail_supported': (_i32, [_i32, _i32, _i32, _i32]),
    'p3d_stem_tail_fwd': (_i32, [_ptr] * 9 + [_i32, _i32, _i32, _i32, _f32, _f32, _ptr, _sz, _ptr]),
    'p3d_stem_tail_bwd': (_i32, [_ptr] * 10 + [_i32, _i32, _i32, _i32, _i32, _ptr, _sz, _ptr]),
    'p3d_softargmax3d_fwd': (_i32, [_ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _f32, _ptr]),
    'p3d_softargmax3d_bwd': (_i32, [_ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _f32, _ptr]),
    'p3d_pose_loss_fwd_bwd': (_i32, [_ptr] * 6 + [_i32, _i32, _i32, _f32, _i32, _f32, _ptr, _ptr]),
    'p3d_masked_loss_fwd_bwd': (_i32, [_ptr] * 5 + [_i32, _i32, _i32, _ptr, _ptr]),
    'p3d_recon_cam_fwd': (_i32, [_ptr] * 4 + [_i32, _i32, _ptr]),
    'p3d_recon_cam_bwd': (_i32, [_ptr] * 6 + [_i32, _i32, _ptr]),
    'p3d_l2norm_sq_accum': (_i32, [_ptr, _i64, _ptr, _ptr]),
    'p3d_adam_step': (_i32, [_ptr, _ptr, _ptr, _ptr, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _ptr, _f32, _ptr]),
    'p3d_adam_step_dev': (_i32, [_ptr, _ptr, _ptr, _ptr, _i64, _f32, _f32, _f32, _f32, _f32, _ptr, _f32, _ptr, _f32, _i32, _ptr, _ptr]),
    'p3d_distill_workspace_bytes': (_sz, [_i32]),
    'p3d_distill_fwd_bwd': (_i32, [_ptr] * 5 + [_i32, _i32, _i32, _i32, _f32, _ptr, _sz, _ptr]),
    'p3d_augment_colour': (_i32, [_ptr, _ptr, _i32, _i32, _i32, _ptr]),
    'p3d_augment_erase': (_i32, [_ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _ptr]),
    'p3d_augment_occlude': (_i32, [_ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _i32, _ptr]),
    'p3d_warp_crops': (_i32, [_ptr, _i32, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _i32, _ptr]),
    'p3d_stream_create': (_i32, [_i32, ctypes.POINTER(ctypes.c_void_p)]),
    'p3d_stream_create_beside': (_i32, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(_i32)]),
    'p3d_stream_create_cumask': (_i32, [ctypes.POINTER(ctypes.c_uint32), _i32, ctypes.POINTER(ctypes.c_void_p)]),
    'p3d_probe_hw_ids': (_i32, [ctypes.c_void_p, ctypes.c_void_p, _i32, _i32]),
    'p3d_stream_destroy': (_i32, [ctypes.c_void_p]),
    'p3d_x3_enable': (_i32, [_i32]),
    'p3d_fx_tune': (None, [_i32, _i32]),
    'p3d_conv_path_stats': (None, [ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double), _i32]),
    'p3d_reproject_crops': (_i32, [_ptr, _i32, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _ptr]),
    'p3d_enhance_depth': (_i32, [_ptr, _ptr, _i64, _f32, _i32, _ptr]),
    'p3d_normalize_rgb': (_i32, [_ptr, _i32, _i32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), _ptr]),
    'p3d_hconv2d_fwd': (_i32, [_desc, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'p3d_hscale_pixels': (_i32, [_ptr, _ptr, _ptr, _i64, _i32, _ptr]),
    'p3d_hconv2d_dgrad': (_i32, [_desc, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'p3d_hblock_fuse_sums': (_i32, [_i32]),
    'p3d_hconv2d_sum_rows': (_i32, [_desc, _i32]),
    'p3d_hconv2d_fwd_stats': (_i32, [_desc, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'p3d_hconv2d_dgrad_sums': (_i32, [_desc, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'p3d_hconv2d_wgrad_workspace_bytes': (_sz, [_desc]),
    'p3d_hconv2d_wgrad': (_i32, [_desc, _ptr, _ptr, _ptr, _ptr, _i32, _f32, _ptr, _sz, _ptr]),
    'p3d_hconv2d_bgrad': (_i32, [_ptr, _i32, _i32, _ptr, _f32, _i32, _ptr]),
    'p3d_nchw_f32_to_nhwc_f16': (_i32, [_ptr, _ptr, _i32, _i32, _i32, _i32, _f32, _ptr]),
    'p3d_nhwc_f16_to_nchw_f32': (_i32, [_ptr, _ptr, _i32, _i32, _i32, _f32, _ptr]),
    'p3d_weight_images_f16': (_i32, [_ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _ptr]),
    'p3d_weight_images_f16_batched': (_i32, [_ptr, _ptr, _ptr, _i32, _ptr]),
    'p3d_hbn_workspace_bytes': (_sz, [_i32]),
    'p3d_hbn_train_fwd': (_i32, [_ptr] * 8 + [_i32, _i32, _f32, _f32, _i32, _ptr, _sz, _ptr]),
    'p3d_hbn_eval_fwd': (_i32, [_ptr] * 7 + [_i32, _i32, _f32, _i32, _ptr, _sz, _ptr]),
    'p3d_hbn_train_bwd': (_i32, [_ptr] * 8 + [_i32, _i32, _i32, _i32, _ptr, _sz, _ptr]),
    'p3d_hbn_frozen_bwd': (_i32, [_ptr] * 8 + [_i32, _i32, _i32, _i32, _ptr, _sz, _ptr]),
    'p3d_hbn_train_fwd_partial': (_i32, [_ptr] * 8 + [_i32, _i32, _f32, _f32, _i32, _ptr, _i32, _ptr, _ptr]),
    'p3d_hbn_train_bwd_mask': (_i32, [_ptr] * 8 + [_i32, _i32, _i32, _ptr, _sz, _ptr]),
    'p3d_hbn_train_bwd_partial': (_i32, [_ptr] * 6 + [_i32, _i32, _i32, _ptr, _i32, _ptr, _ptr]),
    'p3d_hbn_eval_coef': (_i32, [_ptr] * 5 + [_i32, _f32, _ptr]),
    'p3d_hconcat': (_i32, [_ptr, _ptr, _ptr, _i64, _i32, _i32, _i32, _ptr]),
    'p3d_hrelu': (_i32, [_ptr, _ptr, _ptr, _i64, _ptr]),
    'p3d_hmaxpool3x3s2_fwd': (_i32, [_ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _ptr]),
    'p3d_hmaxpool3x3s2_bwd': (_i32, [_ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _ptr]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle.  Raises P3DError if the library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise P3DError('libp3d_hip.so is not built (%s missing): run __graft_entry__.build() or `make -C %s`'
                           % (LIB_PATH, os.path.dirname(LIB_PATH)))
        # PyTorch-ROCm ships its own libamdhip64; it must be the HIP runtime in the process BEFORE this library is mapped, or
        # the kernels would be launched through a second runtime instance that owns no device ("no ROCm-capable device").
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().p3d_last_error()
        raise P3DError('%s failed (%d): %s' % (what, code, msg.decode() if msg else ''))
