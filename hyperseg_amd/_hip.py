"""ctypes binding of libhyperseg_hip.so (include/hyperseg_hip.h).

The library is the product: importing this module without the built .so raises, and every
wrapper refuses non-CUDA / non-fp32 / non-contiguous tensors instead of falling back to anything.
PyTorch is used only for device memory and the current HIP stream.
"""
import ctypes as C
import os

import torch

_LIB_PATH = os.environ.get('HS_HIP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib',
                                                         'libhyperseg_hip.so')   # HS_HIP_LIB: dev override

ACT_NONE, ACT_RELU, ACT_RELU6, ACT_SWISH = 0, 1, 2, 3
PAD_MODES = {'zeros': 0, 'reflect': 1, 'replicate': 2, 'circular': 3}
PREV_NONE, PREV_SAME, PREV_BILINEAR = 0, 1, 2

_STATUS = {-1: 'bad argument', -2: 'feature map does not tile into the weight grid (H % fh or W % fw != 0)',
           -3: 'unsupported shape', -4: 'tile does not fit the LDS'}


class HipLibraryError(RuntimeError):
    pass


class StageInputC(C.Structure):
    _fields_ = [('skip', C.c_void_p), ('prev', C.c_void_p),
                ('batch', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('c_skip', C.c_int32), ('c_prev', C.c_int32),
                ('Hp', C.c_int32), ('Wp', C.c_int32),
                ('coords', C.c_int32), ('prev_mode', C.c_int32)]


class EpilogueC(C.Structure):
    _fields_ = [('scale', C.c_void_p), ('shift', C.c_void_p), ('act', C.c_int32)]


class S2wLayerC(C.Structure):
    _fields_ = [('signal_index', C.c_int32), ('signal_channels', C.c_int32), ('groups', C.c_int32),
                ('wsw_t', C.c_void_p), ('wc', C.c_int32), ('rows', C.c_int32),
                ('bank', C.c_void_p), ('ld', C.c_int64), ('wsw_blk', C.c_void_p)]


class K1LevelC(C.Structure):
    # hs_k1_level of include/hyperseg_hip.h (natural C layout: ctypes pads exactly as the compiler does)
    _fields_ = [('skip', C.c_void_p), ('c_skip', C.c_int32), ('bank', C.c_void_p), ('ld', C.c_int64), ('c_out', C.c_int32),
                ('scale', C.c_void_p), ('shift', C.c_void_p), ('act', C.c_int32)]


class SeTailC(C.Structure):
    # hs_se_tail of include/hyperseg_hip.h
    _fields_ = [('w_reduce', C.c_void_p), ('b_reduce', C.c_void_p), ('w_expand_t', C.c_void_p), ('b_expand', C.c_void_p), ('c_squeezed', C.c_int32),
                ('gate', C.c_void_p), ('squeezed', C.c_void_p), ('workspace', C.c_void_p)]


class ChainIrLevelC(C.Structure):
    # hs_chain_ir_level of include/hyperseg_hip.h
    _fields_ = [('skip', C.c_void_p), ('c_skip', C.c_int32), ('bank', C.c_void_p), ('ld', C.c_int64), ('hidden', C.c_int32), ('c_out', C.c_int32),
                ('s1', C.c_void_p), ('b1', C.c_void_p), ('s2', C.c_void_p), ('b2', C.c_void_p), ('s3', C.c_void_p), ('b3', C.c_void_p)]


class S2wTrainLayerC(C.Structure):
    _fields_ = [('signal_index', C.c_int32), ('signal_channels', C.c_int32), ('groups', C.c_int32),
                ('w', C.c_void_p), ('wc', C.c_int32), ('rows', C.c_int32),
                ('bank', C.c_void_p), ('ld', C.c_int64), ('dbank', C.c_void_p), ('dw', C.c_void_p), ('ds', C.c_void_p)]


def _load():
    if not os.path.exists(_LIB_PATH):
        raise HipLibraryError(
            f'{_LIB_PATH} is missing: build it with `python -m hyperseg_amd.build` (hipcc, gfx950). '
            'hyperseg_amd has no fallback path.')
    lib = C.CDLL(_LIB_PATH)
    i32, i64, vp = C.c_int32, C.c_int64, C.c_void_p
    sig = {
        'hs_version': ([], C.c_int),
        'hs_build_info': ([], C.c_char_p),
        'hs_signal2weights_fwd': ([vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp, i64, vp], C.c_int),
        'hs_signal2weights_multi_fwd': ([vp, i32, i32, i32, i32, C.POINTER(S2wLayerC), i32, vp], C.c_int),
        'hs_s2w_pack_floats': ([i32, i32, i32], C.c_int64),
        'hs_s2w_pack_fwd': ([vp, i32, i32, i32, vp, vp], C.c_int),
        'hs_bank_pack_fwd': ([vp, i32, i32, i32, i32, i32, i32, vp, i64, vp], C.c_int),
        'hs_bn_fold_fwd': ([vp, vp, vp, vp, C.c_float, i32, vp, vp, vp], C.c_int),
        'hs_patch_conv_fwd': ([C.POINTER(StageInputC), i32, i32, vp, i64, i32, i32, i32, i32, i32,
                               C.POINTER(EpilogueC), vp, vp], C.c_int),
        'hs_patch_conv_s2w_fwd': ([C.POINTER(StageInputC), i32, i32, vp, i64, i32, i32, C.POINTER(EpilogueC), vp,
                                   vp, i32, i32, i32, i32, C.POINTER(S2wLayerC), i32, vp], C.c_int),
        'hs_k1_chain_workspace': ([i32, i32, i32, C.POINTER(K1LevelC), i32], C.c_int64),
        'hs_k1_chain_fwd': ([i32, i32, i32, C.POINTER(K1LevelC), i32, vp, vp, vp], C.c_int),
        'hs_decoder_chain_workspace': ([i32, i32, i32, C.POINTER(K1LevelC), i32, C.POINTER(ChainIrLevelC)], C.c_int64),
        'hs_decoder_chain_fwd': ([i32, i32, i32, C.POINTER(K1LevelC), i32, C.POINTER(ChainIrLevelC), vp, vp, vp], C.c_int),
        'hs_meta_conv_fwd': ([vp, i32, i32, i32, i32, vp, i64] + [i32] * 13 + [C.POINTER(EpilogueC), vp, vp], C.c_int),
        'hs_meta_conv_bwd': ([vp, i32, i32, i32, i32, vp, i64] + [i32] * 12 + [vp, vp, vp, i64, vp], C.c_int),
        'hs_patch_conv_gen_fwd': ([C.POINTER(StageInputC), i32, i32, vp, i32, C.POINTER(S2wLayerC), i32,
                                   C.POINTER(EpilogueC), vp, vp], C.c_int),
        'hs_patch_ir_fwd': ([C.POINTER(StageInputC), i32, i32, vp, i64, i32, i32, C.POINTER(EpilogueC),
                             C.POINTER(EpilogueC), C.POINTER(EpilogueC), i32, i32, vp, vp], C.c_int),
        'hs_patch_ir_v0_fwd': ([C.POINTER(StageInputC), i32, i32, vp, i64, i32, i32, C.POINTER(EpilogueC),
                                C.POINTER(EpilogueC), C.POINTER(EpilogueC), i32, vp, vp], C.c_int),
        'hs_patch_ir_v0_ws_fwd': ([C.POINTER(StageInputC), i32, i32, vp, i64, i32, i32, C.POINTER(EpilogueC),
                                   C.POINTER(EpilogueC), C.POINTER(EpilogueC), i32, vp, i64, vp, vp], C.c_int),
        'hs_patch_ir_v0_workspace': ([C.POINTER(StageInputC), i32, i32, i32, i32], C.c_int64),
        'hs_halo_tiles_fwd': ([i32, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp], C.c_int),
        'hs_halo_tiles_bwd': ([i32, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp], C.c_int),
        'hs_tile_interior_fwd': ([i32, vp, i32, i32, i32, i32, i32, i32, vp, vp], C.c_int),
        'hs_tile_interior_bwd': ([i32, vp, i32, i32, i32, i32, i32, i32, vp, vp], C.c_int),
        'hs_dw_tiles_fwd': ([i32, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp, i32, vp], C.c_int),
        'hs_dw_tiles_bwd_in': ([i32, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp, i32, vp], C.c_int),
        'hs_dw_tiles_bwd_w': ([i32, vp, vp, i32, i32, i32, i32, i32, i32, vp, i64, i32, vp], C.c_int),
        'hs_bank_unpack_fwd': ([vp, i64, i32, i32, i32, i32, i32, i32, vp, vp], C.c_int),
        'hs_upsample_bilinear_bwd': ([vp, i64, i32, i32, i32, i32, i32, i32, vp, vp], C.c_int),
        'hs_upsample_bilinear_bf16_fwd': ([vp, i32, i32, i32, i32, i32, i32, vp, vp], C.c_int),
        'hs_upsample_bilinear_typed_bwd': ([i32, vp, i64, i32, i32, i32, i32, i32, i32, vp, vp], C.c_int),
        'hs_bn_train_workspace': ([i32], C.c_int64),
        'hs_bn_train_stats_fwd': ([i32, vp, i32, i32, i64, vp, vp], C.c_int),
        'hs_dw_tiles_bn_fwd': ([i32, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, i32, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp, i32, vp], C.c_int),
        'hs_dw_tiles_bn_bwd_w': ([i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, i64, i32, vp], C.c_int),
        'hs_s2w_train_fwd': ([vp, i32, i32, i32, i32, C.POINTER(S2wTrainLayerC), i32, vp], C.c_int),
        'hs_s2w_train_workspace': ([i32, i32, i32, C.POINTER(S2wTrainLayerC), i32], C.c_int64),
        'hs_s2w_train_bwd': ([vp, i32, i32, i32, i32, C.POINTER(S2wTrainLayerC), i32, vp, vp, i64, vp], C.c_int),
        'hs_cross_entropy_fwd': ([vp, vp, i32, i32, i64, i64, vp, vp], C.c_int),
        'hs_cross_entropy_bwd': ([vp, vp, i32, i32, i64, i64, vp, vp, vp], C.c_int),
        'hs_cross_entropy_typed_fwd': ([i32, vp, vp, i32, i32, i64, i64, vp, vp], C.c_int),
        'hs_cross_entropy_typed_bwd': ([i32, vp, vp, i32, i32, i64, i64, vp, vp, vp], C.c_int),
        'hs_bootstrapped_ce_fwd': ([i32, vp, vp, i32, i32, i64, i64, i32, C.c_float, vp, vp, vp, vp, vp], C.c_int),
        'hs_bootstrapped_ce_bwd': ([i32, vp, vp, i32, i32, i64, i64, vp, vp, vp, vp, vp], C.c_int),
        'hs_bn_act_train_fwd': ([i32, vp, i32, i32, i64, vp, vp, vp, vp, C.c_float, C.c_float, i32, vp, vp, vp, vp, vp, vp], C.c_int),
        'hs_bn_act_train_bwd': ([i32, vp, vp, i32, i32, i64, vp, vp, vp, vp, C.c_float, i32, vp, vp, vp, vp, vp], C.c_int),
        'hs_dw_tiles_bn_bwd_in_partials': ([i32, i32, i32, i32, i32], C.c_int64),
        'hs_dw_tiles_bn_bwd_in': ([i32, vp, vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp], C.c_int),
        'hs_bn_act_train_bwd_apply': ([i32, vp, vp, i32, i32, i64, vp, vp, vp, vp, i32, vp, i64, vp, vp, vp, vp], C.c_int),
        'hs_bootstrap_mean_workspace': ([], C.c_int64),
        'hs_bootstrap_mean_fwd': ([vp, i32, i32, C.c_float, vp, vp, vp], C.c_int),
        'hs_bootstrap_mean_bwd': ([vp, i32, vp, vp, vp, vp], C.c_int),
        'hs_bootstrap_mean_batched_fwd': ([vp, i32, i32, i32, C.c_float, vp, vp, vp], C.c_int),
        'hs_bootstrap_mean_batched_bwd': ([vp, i32, i32, vp, vp, vp, vp], C.c_int),
        'hs_patch_ir_route': ([C.POINTER(StageInputC), i32, i32, i32, i32, i32, i32], C.c_int),
        'hs_ir_tile_map': ([i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), i32], C.c_int),
        'hs_upsample_bilinear_fwd': ([vp, i32, i32, i32, i32, i32, i32, vp, vp], C.c_int),
        'hs_upsample_argmax_fwd': ([vp, i32, i32, i32, i32, i32, i32, vp, vp], C.c_int),
        'hs_stage_input_fwd': ([C.POINTER(StageInputC), vp, vp], C.c_int),
        'hs_stage_input_typed_fwd': ([C.POINTER(StageInputC), i32, i32, vp, vp], C.c_int),
        'hs_patch_conv_bwd_input': ([vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp], C.c_int),
        'hs_patch_conv_bwd_weight': ([vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64, vp], C.c_int),
        'hs_patch_conv_plain_fwd': ([i32, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp], C.c_int),
        'hs_patch_conv_plain_bwd_in': ([i32, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp], C.c_int),
        'hs_patch_conv_plain_bwd_w': ([i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64, vp], C.c_int),
        'hs_depthwise_conv_fwd': ([vp, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp], C.c_int),
        'hs_pointwise_conv_fwd': ([vp, i32, i32, i32, vp, i32, vp, vp, vp, i32, vp, vp, vp], C.c_int),
        'hs_affine_act_fwd': ([vp, i32, i32, i32, vp, vp, i32, vp, vp, vp], C.c_int),
        'hs_patch_conv_bn_fwd': ([i32, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, i32, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp, vp], C.c_int),
        'hs_patch_conv_bn_bwd_w': ([i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64, vp], C.c_int),
        'hs_bootstrap_mean_of_batch_fwd': ([vp, i32, i32, i32, C.c_float, vp, vp, vp, vp], C.c_int),
        'hs_bootstrap_mean_of_batch_bwd': ([vp, i32, i32, vp, vp, vp, vp], C.c_int),
        'hs_adam_blocks': ([vp, i32], C.c_int64),
        'hs_adam_step': ([vp, vp, vp, vp, vp, i32, vp, C.c_float, C.c_double, C.c_double, C.c_float, C.c_float, i32, i32, vp, vp], C.c_int),
        'hs_depthwise_pool_blocks': ([i32, i32], C.c_int),
        'hs_depthwise_conv_se_fwd': ([vp, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp, C.POINTER(SeTailC), vp], C.c_int),
        'hs_mbconv_expand_dw_se_fwd': ([vp, i32, i32, i32, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, C.POINTER(SeTailC), vp], C.c_int),
        'hs_se_tail_workspace': ([i32, i32, i32, i32, i64], C.c_int64),
        'hs_se_tail_tails': ([i32, i32, i32, i64], C.c_int),
        'hs_mbconv_se_workgroups': ([i32, i32, i32, i32, i32, i32], C.c_int64),
        'hs_stem_conv_fwd': ([vp, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp], C.c_int),
        'hs_mbconv_tiles': ([i32, i32, i32, i32], C.c_int),
        'hs_stem_dw_fwd': ([vp, i32, i32, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp], C.c_int),
        'hs_mbconv_expand_dw_fwd': ([vp, i32, i32, i32, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp], C.c_int),
        'hs_se_gate_fwd': ([vp, i32, i32, i32, C.c_float, vp, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp], C.c_int),
        'hs_gemm_split_kp': ([i32], C.c_int),
        'hs_gemm_split_fwd': ([vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp], C.c_int),
        'hs_gemm_split_conv2x2_fwd': ([vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp], C.c_int),
        'hs_gemm_split_up2_fwd': ([vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp], C.c_int),
        'hs_pooled_shift_fwd': ([vp, i32, C.c_float, vp, vp, vp, i32, i32, vp], C.c_int),
    }
    for name, (argtypes, restype) in sig.items():
        fn = getattr(lib, name)          # AttributeError here = the .so does not match the header
        fn.argtypes, fn.restype = argtypes, restype
    if lib.hs_version() != 1:
        raise HipLibraryError(f'ABI mismatch: library reports version {lib.hs_version()}, binding expects 1')
    return lib


lib = _load()
EXPORTS = ['hs_version', 'hs_build_info', 'hs_signal2weights_fwd', 'hs_signal2weights_multi_fwd', 'hs_s2w_pack_floats', 'hs_s2w_pack_fwd', 'hs_bank_pack_fwd', 'hs_bn_fold_fwd',
           'hs_patch_conv_fwd', 'hs_patch_conv_s2w_fwd', 'hs_k1_chain_workspace', 'hs_k1_chain_fwd', 'hs_decoder_chain_workspace', 'hs_decoder_chain_fwd', 'hs_meta_conv_fwd', 'hs_meta_conv_bwd', 'hs_patch_conv_gen_fwd', 'hs_patch_ir_fwd', 'hs_patch_ir_v0_fwd', 'hs_patch_ir_v0_ws_fwd', 'hs_patch_ir_v0_workspace', 'hs_patch_ir_route', 'hs_ir_tile_map', 'hs_upsample_bilinear_fwd', 'hs_upsample_argmax_fwd',
           'hs_stage_input_fwd', 'hs_depthwise_conv_fwd', 'hs_depthwise_pool_blocks', 'hs_stem_conv_fwd', 'hs_stem_dw_fwd', 'hs_mbconv_tiles', 'hs_mbconv_expand_dw_fwd', 'hs_se_gate_fwd', 'hs_depthwise_conv_se_fwd', 'hs_mbconv_expand_dw_se_fwd', 'hs_se_tail_workspace', 'hs_se_tail_tails', 'hs_mbconv_se_workgroups', 'hs_pointwise_conv_fwd', 'hs_affine_act_fwd', 'hs_gemm_split_kp', 'hs_gemm_split_fwd', 'hs_gemm_split_conv2x2_fwd', 'hs_gemm_split_up2_fwd', 'hs_pooled_shift_fwd', 'hs_patch_conv_bwd_input',
           'hs_patch_conv_bwd_weight', 'hs_halo_tiles_fwd', 'hs_halo_tiles_bwd', 'hs_tile_interior_fwd', 'hs_tile_interior_bwd', 'hs_dw_tiles_fwd', 'hs_dw_tiles_bwd_in', 'hs_dw_tiles_bwd_w',
           'hs_s2w_train_fwd', 'hs_s2w_train_workspace', 'hs_s2w_train_bwd', 'hs_cross_entropy_fwd', 'hs_cross_entropy_bwd', 'hs_cross_entropy_typed_fwd', 'hs_cross_entropy_typed_bwd', 'hs_bootstrapped_ce_fwd', 'hs_bootstrapped_ce_bwd', 'hs_bootstrap_mean_workspace', 'hs_bootstrap_mean_fwd', 'hs_bootstrap_mean_bwd', 'hs_bootstrap_mean_batched_fwd', 'hs_bootstrap_mean_batched_bwd', 'hs_bn_train_workspace', 'hs_bn_train_stats_fwd', 'hs_dw_tiles_bn_fwd', 'hs_dw_tiles_bn_bwd_w', 'hs_patch_conv_bn_fwd', 'hs_patch_conv_bn_bwd_w', 'hs_adam_blocks', 'hs_adam_step', 'hs_bootstrap_mean_of_batch_fwd', 'hs_bootstrap_mean_of_batch_bwd', 'hs_upsample_bilinear_bwd', 'hs_upsample_bilinear_typed_bwd', 'hs_upsample_bilinear_bf16_fwd', 'hs_stage_input_typed_fwd', 'hs_bank_unpack_fwd', 'hs_bn_act_train_fwd',
           'hs_bn_act_train_bwd', 'hs_dw_tiles_bn_bwd_in_partials', 'hs_dw_tiles_bn_bwd_in', 'hs_bn_act_train_bwd_apply', 'hs_patch_conv_plain_fwd', 'hs_patch_conv_plain_bwd_in', 'hs_patch_conv_plain_bwd_w']


def check(status, what):
    if status == 0:
        return
    if status < 0:
        raise HipLibraryError(f'{what}: {_STATUS.get(status, status)}')
    raise HipLibraryError(f'{what}: HIP launch failed with hipError_t {status}')


def dev_ptr(t, name='tensor', dtype=torch.float32):
    """Device pointer of a tensor the kernels can consume as-is; raises otherwise (no silent copies)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'{name} must be a torch.Tensor')
    if not t.is_cuda:
        raise HipLibraryError(f'{name} is on {t.device}: the HyperSeg decoder path runs on an MI355X only '
                              '(hyperseg_amd has no CPU fallback)')
    if t.dtype != dtype:
        raise HipLibraryError(f'{name} must be {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise HipLibraryError(f'{name} must be contiguous')
    return t.data_ptr()


class _NullScope:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL_SCOPE = _NullScope()


def device_scope(device):
    """``torch.cuda.device(device)`` only when ``device`` is not already current: the guard's enter / exit is a few microseconds of
    host time per launch, and an eager training step is ~150 launches that are host-bound (the common case: one GPU per process)."""
    if device.index is None or device.index == torch.cuda.current_device():
        return _NULL_SCOPE
    return torch.cuda.device(device)


def stream_ptr(device=None):
    """The caller's current HIP stream on ``device`` (default: the current device).  hyperseg_amd.functional enters
    ``torch.cuda.device(tensor.device)`` around every launch, so 'current' is the device that owns the operands."""
    return torch.cuda.current_stream(device).cuda_stream
