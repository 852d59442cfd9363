"""ctypes binding of libusot_hip.so (the C ABI declared in include/usot_hip.h).

There is no fallback: if the shared library is missing or an entry point fails, the call
raises.  torch is used only for device memory and the current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libusot_hip.so')

ACT_NONE, ACT_RELU, ACT_EXP, ACT_CONF = 0, 1, 2, 3

EXPORTS = (
    'usot_abi_version', 'usot_device_guard', 'usot_device_slot', 'usot_strerror', 'usot_conv2d_f32', 'usot_conv_tile_count',
    'usot_conv_tile_info', 'usot_conv_tile_built', 'usot_experiments_built', 'usot_conv_bf16_tile_built', 'usot_conv_tile_name', 'usot_conv_tile_wfrag', 'usot_conv_tile_xsplit', 'usot_conv_tile_kreq', 'usot_conv_tile_streamk', 'usot_conv_streamk_ws_floats', 'usot_conv_pack_wfrag_f32', 'usot_conv_ws_floats', 'usot_stem_conv_f32', 'usot_maxpool3x3s2_f32',
    'usot_xcorr_depthwise_f32', 'usot_groupdw_f32', 'usot_conf_fusion_reduce_f32',
    'usot_prroi_pool_forward_f32', 'usot_prroi_pool_backward_f32', 'usot_prroi_pool_coor_backward_f32', 'usot_permute4_f32', 'usot_decode_f32',
    'usot_plan_create', 'usot_plan_destroy', 'usot_plan_add_conv', 'usot_plan_add_stem',
    'usot_plan_add_maxpool', 'usot_plan_add_groupdw', 'usot_plan_add_conf_reduce',
    'usot_plan_add_prroi', 'usot_plan_add_permute', 'usot_plan_add_decode', 'usot_plan_run',
    'usot_plan_fork', 'usot_plan_join', 'usot_plan_capture', 'usot_plan_size',
    'usot_groupdw_multi_f32', 'usot_plan_add_groupdw_multi', 'usot_groupdw_multi_lp', 'usot_plan_add_groupdw_multi_lp', 'usot_conf_fusion_reduce_lp', 'usot_plan_add_conf_reduce_lp', 'usot_conv2d_batch_f32', 'usot_plan_add_conv_batch', 'usot_conv2d_bf16', 'usot_conv_bf16_tile_count', 'usot_cvt_f32_to_bf16', 'usot_maxpool3x3s2_bf16',
    'usot_plan_add_conv_bf16', 'usot_plan_add_cvt_bf16', 'usot_plan_add_maxpool_bf16',
    'usot_conv2d_lp', 'usot_cvt_f32_to_lp', 'usot_maxpool3x3s2_lp', 'usot_plan_add_conv_lp', 'usot_plan_add_cvt_lp',
    'usot_plan_add_maxpool_lp', 'usot_stem_pool_lp', 'usot_plan_add_stem_pool_lp',
    'usot_rows_copy_multi_f32', 'usot_plan_add_rows_copy_multi', 'usot_rows_append_gather_f32', 'usot_plan_add_rows_append_gather', 'usot_thin_conv3x3_f32', 'usot_plan_add_thin_conv', 'usot_stem_pool_f32', 'usot_plan_add_stem_pool',
    'usot_plan_profile', 'usot_plan_op_info', 'usot_rows_copy_f32', 'usot_plan_add_rows_copy', 'usot_crop_resize_u8_f32', 'usot_conv_resolve_tile', 'usot_decode_dev_f32',
    'PrRoIPoolingForwardGpu', 'usot_groupdw_auto_variant',
    'usot_stem_pool_ind_f32', 'usot_plan_add_stem_pool_ind', 'usot_bw_probe', 'usot_conv_kstream_lp', 'usot_conv_kstream_supported', 'usot_plan_add_conv_kstream', 'usot_pw_kstream_lp', 'usot_pw_kstream_supported', 'usot_plan_add_pw_kstream', 'usot_conv3x3_halo_lp', 'usot_conv3x3_halo_supported', 'usot_plan_add_conv3x3_halo', 'usot_bneck_first_lp', 'usot_bneck_first_supported', 'usot_plan_add_bneck_first', 'usot_bneck_tail_lp', 'usot_bneck_tail_supported', 'usot_plan_add_bneck_tail', 'usot_pw_panel_lp', 'usot_pw_panel_supported', 'usot_pw_panel_pixels', 'usot_pw_panel_min_pixels', 'usot_plan_add_pw_panel', 'usot_pw_panel_pair_lp', 'usot_pw_panel_pair_supported', 'usot_plan_add_pw_panel_pair', 'usot_conv_pw_lp', 'usot_conv_pw_supported', 'usot_conv_pw_pixels', 'usot_plan_add_conv_pw', 'usot_conv_pw_pair_lp', 'usot_conv_pw_pair_supported', 'usot_plan_add_conv_pw_pair', 'usot_conv_pw_ov_lp', 'usot_conv_pw_ov_supported', 'usot_conv_pw_ov_ws_bytes', 'usot_plan_add_conv_pw_ov', 'usot_conv_pw_ov_trace', 'usot_stem_conv_mu_f32', 'usot_stem_pool_mu_f32', 'usot_plan_add_stem_pool_mu', 'usot_plan_add_stem_mu', 'usot_pw_pair_lp', 'usot_pw_pair_layout', 'usot_pw_pair_supported', 'usot_plan_add_pw_pair', 'usot_pw_pair_f32', 'usot_pw_pair_f32s', 'usot_pw_pair_f32s_supported', 'usot_pw_pair_f32_supported', 'usot_pw_pair_f32_ws_floats', 'usot_pw_single_f32', 'usot_pw_single_f32_supported', 'usot_plan_add_pw_single', 'usot_stream_conv3x3_f32', 'usot_stream_conv3x3_f32_supported', 'usot_plan_add_stream_conv3x3', 'usot_pw_triple_f32', 'usot_pw_triple_f32_supported', 'usot_plan_add_pw_triple',
)


class HipError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w', C.c_void_p), ('bias', C.c_void_p), ('res', C.c_void_p),
                ('y', C.c_void_p), ('ws', C.c_void_p),
                ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('Cin', C.c_int32),
                ('OH', C.c_int32), ('OW', C.c_int32), ('Cout', C.c_int32),
                ('KH', C.c_int32), ('KW', C.c_int32), ('stride', C.c_int32),
                ('pad_h', C.c_int32), ('pad_w', C.c_int32), ('dil_h', C.c_int32), ('dil_w', C.c_int32),
                ('y_cstride', C.c_int32), ('y_coff', C.c_int32), ('res_cstride', C.c_int32),
                ('res_coff', C.c_int32), ('y_nchw', C.c_int32),
                ('act', C.c_int32), ('act2', C.c_int32), ('act_split', C.c_int32),
                ('groups', C.c_int32),
                ('x_gs', C.c_int64), ('w_gs', C.c_int64), ('b_gs', C.c_int64), ('y_gs', C.c_int64),
                ('r_gs', C.c_int64),
                ('ksplit', C.c_int32), ('tile', C.c_int32), ('w_frag', C.c_int32), ('defer', C.c_int32),
                ('w_scale', C.c_void_p), ('x_split', C.c_int32), ('y_split', C.c_int32), ('ovf', C.c_void_p)]


class GroupDWDesc(C.Structure):
    _fields_ = [('x', C.c_void_p * 3), ('z', C.c_void_p * 3), ('out', C.c_void_p),
                ('hk', C.c_int32 * 3), ('wk', C.c_int32 * 3),
                ('x_cs', C.c_int32 * 3), ('x_co', C.c_int32 * 3),
                ('z_cs', C.c_int32 * 3), ('z_co', C.c_int32 * 3),
                ('wsm', C.c_float * 3),
                ('S', C.c_int32), ('x_rep', C.c_int32), ('OH', C.c_int32), ('OW', C.c_int32),
                ('C', C.c_int32), ('cols_per_thread', C.c_int32)]


class PwPairDesc(C.Structure):
    _fields_ = [('t2', C.c_void_p), ('w3p', C.c_void_p), ('res', C.c_void_p), ('w1', C.c_void_p),
                ('b3', C.c_void_p), ('b1', C.c_void_p), ('y', C.c_void_p), ('t', C.c_void_p),
                ('M', C.c_int32), ('CM', C.c_int32), ('CO', C.c_int32), ('CN', C.c_int32), ('act2', C.c_int32),
                ('ws', C.c_void_p), ('t2_parts', C.c_int32), ('res_parts', C.c_int32), ('t2_bias', C.c_void_p), ('res_bias', C.c_void_p),
                ('ovf', C.c_void_p)]


class BneckDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w1', C.c_void_p), ('w2', C.c_void_p), ('w3c', C.c_void_p), ('wn', C.c_void_p),
                ('b1', C.c_void_p), ('b2', C.c_void_p), ('b3c', C.c_void_p), ('bn', C.c_void_p),
                ('y', C.c_void_p), ('t', C.c_void_p), ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32)]


_lib = None


def lib():
    """The loaded library; raises HipError when it has not been built."""
    global _lib, LIB_PATH
    if _lib is None:
        LIB_PATH = os.environ.get('USOT_HIP_LIB', LIB_PATH)      # e.g. the -DUSOT_TRACE build of scripts/trace_kstep.py
        if not os.path.exists(LIB_PATH):
            raise HipError('%s is missing: run `python -m usot_amd.build` (hipcc, gfx950). '
                           'There is no CPU fallback for the tracking forward pass.' % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.usot_strerror.restype = C.c_char_p
        L.usot_strerror.argtypes = [C.c_int]
        L.usot_conv_ws_floats.restype = C.c_int64
        L.usot_plan_create.restype = C.c_void_p
        L.usot_plan_destroy.argtypes = [C.c_void_p]
        for name in ('usot_plan_add_conv', 'usot_plan_add_groupdw'):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
        L.usot_plan_add_stem.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int] * 5
        L.usot_plan_add_maxpool.argtypes = [C.c_void_p] + [C.c_void_p] * 2 + [C.c_int] * 6
        L.usot_plan_add_conf_reduce.argtypes = [C.c_void_p] + [C.c_void_p] * 2 + [C.c_int] * 4
        L.usot_plan_add_prroi.argtypes = ([C.c_void_p] + [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_float]
                                          + [C.c_int64] * 8)
        L.usot_plan_add_permute.argtypes = [C.c_void_p] + [C.c_void_p] * 2 + [C.c_int] * 4 + [C.c_int64] * 4
        L.usot_plan_add_decode.argtypes = ([C.c_void_p] + [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_float]
                                           + [C.c_double] * 2 + [C.c_void_p] * 2)
        L.usot_plan_add_rows_copy.argtypes = [C.c_void_p] + [C.c_void_p] * 3 + [C.c_int] * 3
        L.usot_rows_copy_f32.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3
        L.usot_thin_conv3x3_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.usot_stem_pool_f32.argtypes = [C.c_void_p] * 5 + [C.c_int] * 7
        L.usot_plan_add_stem_pool.argtypes = [C.c_void_p] * 5 + [C.c_int] * 7
        L.usot_plan_add_stem_pool_ind.argtypes = [C.c_void_p] * 5 + [C.c_int] * 7 + [C.c_void_p]
        L.usot_stem_pool_ind_f32.argtypes = [C.c_void_p] * 5 + [C.c_int] * 7 + [C.c_void_p]
        L.usot_conv3x3_halo_lp.argtypes = [C.c_void_p] * 5 + [C.c_int] * 7
        L.usot_plan_add_conv3x3_halo.argtypes = [C.c_void_p] * 5 + [C.c_int] * 7
        L.usot_conv3x3_halo_supported.argtypes = [C.c_int] * 2
        L.usot_bneck_first_lp.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.usot_plan_add_bneck_first.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.usot_bneck_first_supported.argtypes = [C.c_int] * 4
        L.usot_bneck_tail_lp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.usot_plan_add_bneck_tail.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.usot_bneck_tail_supported.argtypes = [C.c_int] * 3
        L.usot_bw_probe.argtypes = [C.c_void_p] * 3 + [C.c_int64, C.c_int]
        L.usot_conv_kstream_lp.argtypes = [C.c_void_p] * 5 + [C.c_int] * 10
        L.usot_plan_add_conv_kstream.argtypes = [C.c_void_p] * 5 + [C.c_int] * 10
        L.usot_conv_kstream_supported.argtypes = [C.c_int] * 4
        L.usot_pw_kstream_lp.argtypes = [C.c_void_p] * 5 + [C.c_long] + [C.c_int] * 4
        L.usot_plan_add_pw_kstream.argtypes = [C.c_void_p] * 5 + [C.c_long] + [C.c_int] * 4
        L.usot_pw_kstream_supported.argtypes = [C.c_int] * 2
        L.usot_pw_panel_lp.argtypes = [C.c_void_p] * 6 + [C.c_int] * 5
        L.usot_plan_add_pw_panel.argtypes = [C.c_void_p] * 6 + [C.c_int] * 5
        L.usot_pw_panel_supported.argtypes = [C.c_int] * 2
        L.usot_pw_panel_pair_supported.argtypes = [C.c_int] * 3
        L.usot_pw_panel_pixels.argtypes = [C.c_int] * 3
        L.usot_pw_panel_min_pixels.argtypes = [C.c_int] * 2
        L.usot_pw_panel_pair_lp.argtypes = [C.c_void_p, C.POINTER(PwPairDesc), C.c_int]
        L.usot_plan_add_pw_panel_pair.argtypes = [C.c_void_p, C.POINTER(PwPairDesc), C.c_int]
        L.usot_conv_pw_lp.argtypes = [C.c_void_p] * 6 + [C.c_int]
        L.usot_plan_add_conv_pw.argtypes = [C.c_void_p] * 6 + [C.c_int]
        L.usot_conv_pw_supported.argtypes = [C.c_int] * 3
        L.usot_conv_pw_pixels.argtypes = [C.c_int64]
        L.usot_conv_pw_pair_lp.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(PwPairDesc), C.c_int]
        L.usot_plan_add_conv_pw_pair.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(PwPairDesc), C.c_int]
        L.usot_conv_pw_pair_supported.argtypes = [C.c_int] * 3
        L.usot_conv_pw_ov_lp.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(PwPairDesc), C.c_int, C.c_void_p]
        L.usot_plan_add_conv_pw_ov.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(PwPairDesc), C.c_int, C.c_void_p]
        L.usot_conv_pw_ov_supported.argtypes = [C.c_int] * 3
        L.usot_conv_pw_ov_ws_bytes.argtypes = [C.c_int64]
        L.usot_conv_pw_ov_ws_bytes.restype = C.c_int64
        L.usot_conv_pw_ov_trace.argtypes = [C.c_void_p]
        L.usot_stem_pool_mu_f32.argtypes = [C.c_void_p] * 5 + [C.c_int] * 7 + [C.c_void_p] + [C.c_float] * 3
        L.usot_plan_add_stem_pool_mu.argtypes = [C.c_void_p] * 5 + [C.c_int] * 7 + [C.c_void_p] + [C.c_float] * 3
        L.usot_plan_add_stem_mu.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_float] * 3
        L.usot_stem_conv_mu_f32.argtypes = [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_float] * 3
        L.usot_plan_add_thin_conv.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.usot_rows_copy_multi_f32.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.usot_plan_add_rows_copy_multi.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.usot_rows_append_gather_f32.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int]
        L.usot_plan_add_rows_append_gather.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int]
        L.usot_crop_resize_u8_f32.argtypes = [C.c_void_p] * 3 + [C.c_int] * 9
        L.usot_plan_add_conv_bf16.argtypes = [C.c_void_p, C.c_void_p]
        L.usot_plan_add_conv_lp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.usot_pw_pair_lp.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.usot_plan_add_pw_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.usot_pw_pair_layout.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_int32)] * 2
        L.usot_pw_pair_supported.argtypes = [C.c_int] * 3
        L.usot_pw_pair_f32.argtypes = [C.c_void_p, C.c_void_p]
        L.usot_pw_pair_f32_supported.argtypes = [C.c_int] * 3
        L.usot_pw_pair_f32s.argtypes = [C.c_void_p, C.c_void_p]
        L.usot_pw_pair_f32s_supported.argtypes = [C.c_int] * 3
        L.usot_pw_pair_f32_ws_floats.argtypes = [C.c_int] * 4
        L.usot_pw_pair_f32_ws_floats.restype = C.c_int64
        L.usot_pw_single_f32.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4
        L.usot_pw_single_f32_supported.argtypes = [C.c_int] * 2
        L.usot_plan_add_pw_single.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4
        L.usot_stream_conv3x3_f32.argtypes = [C.c_void_p] * 6 + [C.c_int] * 12
        L.usot_plan_add_stream_conv3x3.argtypes = [C.c_void_p] * 6 + [C.c_int] * 12
        L.usot_stream_conv3x3_f32_supported.argtypes = [C.c_int] * 2
        L.usot_pw_triple_f32.argtypes = [C.c_void_p] * 5 + [C.c_int] * 10
        L.usot_plan_add_pw_triple.argtypes = [C.c_void_p] * 5 + [C.c_int] * 10
        L.usot_pw_triple_f32_supported.argtypes = [C.c_int] * 4
        L.usot_plan_add_cvt_lp.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.usot_plan_add_maxpool_lp.argtypes = [C.c_void_p] + [C.c_void_p] * 2 + [C.c_int] * 7
        L.usot_conv2d_lp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.usot_stem_pool_lp.argtypes = [C.c_void_p] * 5 + [C.c_int] * 8 + [C.c_float] * 3
        L.usot_plan_add_stem_pool_lp.argtypes = [C.c_void_p] * 5 + [C.c_int] * 8 + [C.c_float] * 3
        L.usot_plan_add_cvt_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.usot_plan_add_maxpool_bf16.argtypes = [C.c_void_p] + [C.c_void_p] * 2 + [C.c_int] * 6
        L.usot_conv2d_bf16.argtypes = [C.c_void_p, C.c_void_p]
        L.usot_cvt_f32_to_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.usot_maxpool3x3s2_bf16.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6
        L.usot_plan_fork.argtypes = [C.c_void_p, C.c_int]
        L.usot_plan_join.argtypes = [C.c_void_p, C.c_int]
        L.usot_plan_run.argtypes = [C.c_void_p, C.c_void_p]
        L.usot_plan_capture.argtypes = [C.c_void_p, C.c_void_p]
        L.usot_plan_size.argtypes = [C.c_void_p]
        L.usot_plan_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.usot_plan_op_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.usot_conv2d_f32.argtypes = [C.c_void_p, C.c_void_p]
        L.usot_conv2d_batch_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.usot_plan_add_conv_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.usot_groupdw_f32.argtypes = [C.c_void_p, C.c_void_p]
        L.usot_groupdw_multi_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.usot_plan_add_groupdw_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.usot_plan_add_groupdw_multi_lp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.usot_groupdw_multi_lp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.usot_conf_fusion_reduce_lp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 5
        L.usot_plan_add_conf_reduce_lp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 5
        L.usot_stem_conv_f32.argtypes = [C.c_void_p] * 5 + [C.c_int] * 5
        L.usot_maxpool3x3s2_f32.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6
        L.usot_xcorr_depthwise_f32.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5
        L.usot_conf_fusion_reduce_f32.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4
        L.usot_prroi_pool_forward_f32.argtypes = ([C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_float]
                                                  + [C.c_int64] * 8)
        L.PrRoIPoolingForwardGpu.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_float, C.c_int]
        L.PrRoIPoolingForwardGpu.restype = None
        L.usot_prroi_pool_backward_f32.argtypes = [C.c_void_p] * 4 + [C.c_int] * 7 + [C.c_float]
        L.usot_prroi_pool_coor_backward_f32.argtypes = [C.c_void_p] * 6 + [C.c_int] * 7 + [C.c_float]
        for name in ('PrRoIPoolingBackwardGpu', 'PrRoIPoolingCoorBackwardGpu'):
            getattr(L, name).argtypes = [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_float, C.c_int, C.c_int]
            getattr(L, name).restype = None
        L.usot_permute4_f32.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_int64] * 4
        L.usot_decode_f32.argtypes = ([C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_float]
                                      + [C.c_double] * 4)
        L.usot_decode_dev_f32.argtypes = ([C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_float] + [C.c_double] * 2
                                          + [C.c_void_p] * 2)
        L.usot_conv_tile_info.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.usot_conv_pack_wfrag_f32.argtypes = [C.c_void_p] * 3 + [C.c_int] * 2
        L.usot_conv_tile_kreq.argtypes = [C.c_int, C.POINTER(C.c_int)]
        L.usot_conv_streamk_ws_floats.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.usot_conv_streamk_ws_floats.restype = C.c_int64
        _lib = L
    return _lib


def check(code, what=''):
    if code != 0:
        raise HipError('%s failed: %s (%d)' % (what or 'usot call', lib().usot_strerror(code).decode(), code))


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype=torch.float32):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise HipError('expected a device tensor (the HIP path has no CPU implementation)')
    if t.dtype != dtype:
        raise HipError('expected %s, got %s' % (dtype, t.dtype))
    return t


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def tile_name(tile):
    buf = C.create_string_buffer(96)
    check(lib().usot_conv_tile_name(int(tile), buf, 96), 'usot_conv_tile_name')
    return buf.value.decode()


def tile_built(tile):
    """Whether conv tile id `tile` is compiled into the library: the ROUTED tiles always, the experimental ids (lab notebook:
    parity-green, measured slower, never selected by the engine) only when it was built with USOT_EXPERIMENTS=1."""
    return bool(lib().usot_conv_tile_built(int(tile)))


def lp_tile_built(tile):
    return bool(lib().usot_conv_bf16_tile_built(int(tile)))


def experiments_built():
    return bool(lib().usot_experiments_built())


def tile_wfrag(tile):
    """1 when conv tile id `tile` streams its filters in fragment order (descriptor field w_frag, pack_wfrag)."""
    return int(lib().usot_conv_tile_wfrag(int(tile))) if tile else 0


def tile_xsplit(tile):
    """1 when conv tile id `tile` reads its input map in the split-fp16 layout (descriptor field x_split, split_map)."""
    return int(lib().usot_conv_tile_xsplit(int(tile))) if tile else 0


def split_map(x):
    """NHWC fp32 map [..., C] (C % 64 == 0) -> the split-fp16 layout of usot_conv_desc.x_split / y_split as a float32-typed tensor of the
    same shape: per pixel and 64-channel block the 64 hi halves then the 64 lo halves of 8 x value."""
    c = x.shape[-1]
    assert c % 64 == 0
    v = x.float() * SPLIT16_X_SCALE
    hi = v.half()
    lo = (v - hi.float()).half()
    blocks = lambda t: t.reshape(tuple(x.shape[:-1]) + (c // 64, 64))
    return torch.cat([blocks(hi), blocks(lo)], -1).contiguous().view(torch.float32).reshape(x.shape)


def unsplit_map(y):
    """Inverse of split_map (hi + lo, / 8) - for tests and probes."""
    c = y.shape[-1]
    h = y.contiguous().view(torch.float16).reshape(tuple(y.shape[:-1]) + (c // 64, 2, 64)).float()
    return ((h[..., 0, :] + h[..., 1, :]) / SPLIT16_X_SCALE).reshape(y.shape)


def tile_kreq(tile):
    """(K, Cin multiple) a weight-stationary conv tile requires, or (0, 0) for the tiles that take any geometry."""
    if not tile:
        return 0, 0
    kp = C.c_int(0)
    k = int(lib().usot_conv_tile_kreq(int(tile), C.byref(kp)))
    return k, int(kp.value) if k else 0


def tile_supports(tile, cin, cout, k):
    """Whether conv tile id `tile` can run a convolution with these channel counts and reduction length."""
    kreq, kp = tile_kreq(tile)
    if not kreq:
        return True
    return k == kreq and cin % kp == 0 and cout % 32 == 0


def tile_streamk(tile):
    return int(lib().usot_conv_tile_streamk(int(tile))) if tile else 0


def streamk_ws(descs, tile, device):
    """Zeroed workspace of a persistent stream-K launch of `descs` (list of ConvDesc) on `tile`; sets every desc's ws."""
    arr = (ConvDesc * len(descs))(*descs)
    n = int(lib().usot_conv_streamk_ws_floats(arr, len(descs), int(tile)))
    if n < 0:
        raise HipError('usot_conv_streamk_ws_floats: %d' % n)
    ws = torch.zeros(max(n, 4), device=device, dtype=torch.float32)
    for d in descs:
        d.ws = ws.data_ptr()
    return ws


def pack_wfrag(w):
    """[Cout, K] row-major fp32 filter bank (K % 64 == 0) -> the same bank in MFMA fragment order, rows padded to a multiple
    of 16 (usot_conv_pack_wfrag_f32).  Returns a new device tensor [ceil(Cout/16)*16, K]."""
    _dev(w)
    cout, k = w.shape
    wf = torch.empty(((cout + 15) // 16) * 16, k, device=w.device, dtype=torch.float32)
    check(lib().usot_conv_pack_wfrag_f32(stream(), ptr(w.contiguous()), ptr(wf), cout, k), 'usot_conv_pack_wfrag_f32')
    return wf


def tile_table():
    L = lib()
    out = {}
    for i in range(1, L.usot_conv_tile_count() + 1):
        if not L.usot_conv_tile_built(i):
            continue
        bm, bn = C.c_int(), C.c_int()
        L.usot_conv_tile_info(i, C.byref(bm), C.byref(bn))
        out[i] = (bm.value, bn.value)
    return out


# ------------------------------------------------------------------ tensor-level wrappers

SPLIT16_X_SCALE = 8.0      # csrc/conv_igemm.hip (PF = 4): the activations are split as hi + lo fp16 of 8 x


def split16_pack(w):
    """Filter bank [rows][K] fp32 (K % 64 == 0) -> (bank in the split-fp16 layout of usot_conv_desc.w_frag = 2 as a float32-typed
    tensor [rows][K]: per row and k-tile of 64, 64 hi halves then 64 lo halves of row x 2^e, e chosen per row so that the largest
    |value| lands in [512, 1024); the per-row factors 1 / (2^e x SPLIT16_X_SCALE) that undo both scales, float32 [rows])."""
    rows, k = w.shape
    assert k % 64 == 0
    w = w.float()
    amax = w.abs().amax(1)
    e = torch.where(amax > 0, torch.floor(torch.log2(1024.0 / amax.clamp_min(1e-30))), torch.zeros_like(amax))
    e = torch.where(amax * torch.exp2(e) >= 1024.0, e - 1, e).clamp(-100, 110)     # (guards the log2 rounding at exact powers of two)
    sw = torch.exp2(e)
    ws = w * sw[:, None]
    hi = ws.half()
    lo = (ws - hi.float()).half()
    packed = torch.cat([hi.view(rows, k // 64, 64), lo.view(rows, k // 64, 64)], 2).contiguous()      # [rows][KT][128] halves
    return packed.view(torch.float32).view(rows, k).contiguous(), (1.0 / (sw * SPLIT16_X_SCALE)).float().contiguous()


def conv_desc(x, w, bias, y, *, N, H, W, Cin, OH, OW, Cout, KH, KW, stride=1, pad=(0, 0), dil=(1, 1),
              res=None, act=ACT_NONE, act2=ACT_NONE, act_split=0, y_cstride=0, y_coff=0,
              res_cstride=0, res_coff=0, y_nchw=0, groups=1, x_gs=0, w_gs=0, b_gs=0, y_gs=0, r_gs=0,
              ksplit=1, tile=0, ws=None, w_frag=0, defer=0, w_scale=None, x_split=0, y_split=0, ovf=None):
    d = ConvDesc()
    d.x, d.w, d.bias, d.res, d.y, d.ws = (x, w, bias or None, res or None, y, ws or None)
    d.N, d.H, d.W, d.Cin, d.OH, d.OW, d.Cout = N, H, W, Cin, OH, OW, Cout
    d.KH, d.KW, d.stride = KH, KW, stride
    d.pad_h, d.pad_w, d.dil_h, d.dil_w = pad[0], pad[1], dil[0], dil[1]
    d.y_cstride, d.y_coff, d.res_cstride, d.res_coff, d.y_nchw = y_cstride, y_coff, res_cstride, res_coff, y_nchw
    d.act, d.act2, d.act_split = act, act2, act_split
    d.groups = groups
    d.x_gs, d.w_gs, d.b_gs, d.y_gs, d.r_gs = x_gs, w_gs, b_gs, y_gs, r_gs
    d.ksplit, d.tile, d.w_frag, d.defer = ksplit, tile, w_frag, defer
    d.w_scale = w_scale or None
    d.x_split, d.y_split = int(x_split), int(y_split)
    d.ovf = ovf or None          # split-fp16 tiles: the sticky "a finished sum was not finite" word (usot_conv_desc.ovf)
    return d


def conv2d(x, w, bias, *, KH, KW, stride=1, pad=(0, 0), dil=(1, 1), res=None, act=ACT_NONE,
           tile=0, ksplit=1, y_nchw=False, y_split=False, ovf=None):
    """x NHWC [N,H,W,Cin] dense, w packed [Cout, KH*KW*Cin] -> y NHWC [N,OH,OW,Cout].
    ovf: int32 device tensor, the sticky range word of the split-fp16 tiles (usot_conv_desc.ovf)."""
    _dev(x), _dev(w)
    N, H, W_, Cin = x.shape
    Cout = w.shape[0]
    OH = (H + 2 * pad[0] - dil[0] * (KH - 1) - 1) // stride + 1
    OW = (W_ + 2 * pad[1] - dil[1] * (KW - 1) - 1) // stride + 1
    shape = (N, Cout, OH, OW) if y_nchw else (N, OH, OW, Cout)
    y = torch.empty(shape, device=x.device, dtype=torch.float32)
    ws = None
    if ksplit > 1:        # slabs + tile tickets (usot_conv_ws_floats); the tickets must start at zero
        ws = torch.zeros(ksplit * N * OH * OW * Cout + ((N * OH * OW + 15) // 16) * ((Cout + 31) // 32), device=x.device,
                         dtype=torch.float32)
    frag = tile_wfrag(tile)
    wsc = None
    xs = tile_xsplit(tile)
    if xs:                  # all-DMA split-fp16 tile: the input map in the split layout (here: converted for the caller)
        x = split_map(x)
    if frag == 2:           # split-fp16 tile: hi + lo fp16 of every scaled filter row, and the scales
        w, wsc = split16_pack(w)
    elif frag:              # weight-streaming tile: the filter bank in MFMA fragment order
        w = pack_wfrag(w)
    d = conv_desc(x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                  N=N, H=H, W=W_, Cin=Cin, OH=OH, OW=OW, Cout=Cout, KH=KH, KW=KW, stride=stride, pad=pad,
                  dil=dil, res=res.data_ptr() if res is not None else None, act=act, tile=tile,
                  ksplit=ksplit, ws=ws.data_ptr() if ws is not None else None, y_nchw=int(y_nchw), w_frag=frag,
                  w_scale=wsc.data_ptr() if wsc is not None else None, x_split=xs, y_split=int(y_split),
                  ovf=ovf.data_ptr() if ovf is not None else None)
    if tile_streamk(tile):
        keep = streamk_ws([d], tile, x.device)
    check(lib().usot_conv2d_f32(stream(), C.byref(d)), 'usot_conv2d_f32')
    return y


def stem_conv(x, w, bias, mu=(0.0, 0.0, 0.0)):
    """mu: per-input-channel offsets subtracted from the crop while it is staged (bias must carry + sum(w) * mu)."""
    _dev(x), _dev(w), _dev(bias)
    N, c, H, W_ = x.shape
    assert c == 3 and x.is_contiguous()
    OH, OW = (H - 7) // 2 + 1, (W_ - 7) // 2 + 1
    y = torch.empty((N, OH, OW, 64), device=x.device, dtype=torch.float32)
    check(lib().usot_stem_conv_mu_f32(stream(), ptr(x), ptr(w), ptr(bias), ptr(y), N, H, W_, OH, OW, *[float(v) for v in mu]),
          'usot_stem_conv_mu_f32')
    return y


def stem_pool(x, wfrag, bias, mu=(0.0, 0.0, 0.0)):
    """Fused fp32 stem + max-pool: x NCHW fp32 -> NHWC [N][PH][PW][64]; wfrag from engine.pack_stem_f32; mu as in stem_conv."""
    _dev(x), _dev(wfrag), _dev(bias)
    N, c, H, W_ = x.shape
    assert c == 3 and x.is_contiguous()
    OH, OW = (H - 7) // 2 + 1, (W_ - 7) // 2 + 1
    PH, PW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
    y = torch.empty((N, PH, PW, 64), device=x.device, dtype=torch.float32)
    check(lib().usot_stem_pool_mu_f32(stream(), ptr(x), ptr(wfrag), ptr(bias), ptr(y), N, H, W_, OH, OW, PH, PW, None,
                                      *[float(v) for v in mu]), 'usot_stem_pool_mu_f32')
    return y


def stem_pool_lp(x, wfrag, bias, dtype=torch.bfloat16, mu=(0.0, 0.0, 0.0)):
    """Fused low-precision stem + max-pool: x NCHW fp32 -> NHWC [N][PH][PW][64] in `dtype`;
    wfrag from usot_amd.engine.pack_stem_lp; the kernel convolves x - mu[ci] (bias must hold the
    folded mu term).  bf16 output with fp16 fragments = the kernel's mode 2 (fp16 arithmetic, bf16 storage)."""
    _dev(x), _dev(wfrag, wfrag.dtype), _dev(bias)
    if wfrag.dtype not in (torch.float16, torch.bfloat16) or (dtype == torch.float16 and wfrag.dtype != dtype):
        raise HipError('stem_pool_lp: fragments %s for output %s' % (wfrag.dtype, dtype))
    N, c, H, W_ = x.shape
    assert c == 3 and x.is_contiguous()
    OH, OW = (H - 7) // 2 + 1, (W_ - 7) // 2 + 1
    PH, PW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
    y = torch.empty((N, PH, PW, 64), device=x.device, dtype=dtype)
    check(lib().usot_stem_pool_lp(stream(), ptr(x), ptr(wfrag), ptr(bias), ptr(y), N, H, W_, OH, OW, PH, PW,
                                  1 if dtype == torch.float16 else (2 if wfrag.dtype == torch.float16 else 0), float(mu[0]), float(mu[1]), float(mu[2])),
          'usot_stem_pool_lp')
    return y


def maxpool3x3s2(x):
    _dev(x)
    N, H, W_, c = x.shape
    OH, OW = (H - 1) // 2 + 1, (W_ - 1) // 2 + 1
    y = torch.empty((N, OH, OW, c), device=x.device, dtype=torch.float32)
    check(lib().usot_maxpool3x3s2_f32(stream(), ptr(x), ptr(y), N, H, W_, c, OH, OW), 'usot_maxpool3x3s2_f32')
    return y


def xcorr_depthwise(x, kernel):
    """Drop-in for reference connect.py:147-157 on NCHW device tensors."""
    _dev(x), _dev(kernel)
    b, c, hk, wk = kernel.shape
    hx, wx = x.shape[2], x.shape[3]
    x = x.contiguous().view(-1, hx, wx)
    k = kernel.contiguous().view(-1, hk, wk)
    if x.shape[0] != k.shape[0]:
        raise HipError('xcorr_depthwise: plane counts differ (%d vs %d)' % (x.shape[0], k.shape[0]))
    out = torch.empty((b, c, hx - hk + 1, wx - wk + 1), device=x.device, dtype=torch.float32)
    check(lib().usot_xcorr_depthwise_f32(stream(), ptr(x), ptr(k), ptr(out), b * c, hx, wx, hk, wk),
          'usot_xcorr_depthwise_f32')
    return out


def groupdw_desc(xs, zs, out, wsm, *, S, x_rep, OH, OW, Cc, x_cs, x_co, z_cs, z_co, cols=0):
    d = GroupDWDesc()
    geo = ((5, 5), (3, 5), (5, 3))
    for b in range(3):
        d.x[b], d.z[b] = xs[b], zs[b]
        d.hk[b], d.wk[b] = geo[b]
        d.x_cs[b], d.x_co[b], d.z_cs[b], d.z_co[b] = x_cs[b], x_co[b], z_cs[b], z_co[b]
        d.wsm[b] = float(wsm[b])
    d.out = out
    d.S, d.x_rep, d.OH, d.OW, d.C, d.cols_per_thread = S, x_rep, OH, OW, Cc, cols
    return d


def groupdw(xs, zs, wsm, x_rep=1, cols=0):
    """xs: 3 NHWC search maps [XS,h,w,C]; zs: 3 NHWC templates [S,hk,wk,C] -> [S,OH,OW,C]."""
    S, Cc = zs[0].shape[0], zs[0].shape[3]
    OH, OW = xs[0].shape[1] - 4, xs[0].shape[2] - 4
    for t in list(xs) + list(zs):
        _dev(t)
        assert t.is_contiguous()
    out = torch.empty((S, OH, OW, Cc), device=xs[0].device, dtype=torch.float32)
    d = groupdw_desc([t.data_ptr() for t in xs], [t.data_ptr() for t in zs], out.data_ptr(), wsm,
                     S=S, x_rep=x_rep, OH=OH, OW=OW, Cc=Cc, x_cs=[Cc] * 3, x_co=[0] * 3,
                     z_cs=[Cc] * 3, z_co=[0] * 3, cols=cols)
    check(lib().usot_groupdw_f32(stream(), C.byref(d)), 'usot_groupdw_f32')
    return out


GROUPDW_VARIANTS = {1: 'groupdw_nhwc_kernel<5,1> (strips)', 2: 'groupdw_nhwc_col_kernel', 3: 'groupdw_nhwc_stream_kernel',
                    4: 'groupdw_nhwc_ring_kernel', 5: 'groupdw_nhwc_kernel<5,5> (patches)', 6: 'groupdw_dma_kernel'}


def groupdw_variant_name(samples, ow=25):
    """Kernel the launcher's auto mode runs for `samples` samples (what a benchmark times)."""
    return GROUPDW_VARIANTS.get(int(lib().usot_groupdw_auto_variant(int(samples), int(ow))), 'groupdw')


def conf_fusion_reduce(cv, B, M):
    _dev(cv)
    P, C2 = cv.shape[1] * cv.shape[2], cv.shape[3]
    out = torch.empty((B, cv.shape[1], cv.shape[2], C2 // 2), device=cv.device, dtype=torch.float32)
    check(lib().usot_conf_fusion_reduce_f32(stream(), ptr(cv), ptr(out), B, M, P, C2 // 2),
          'usot_conf_fusion_reduce_f32')
    return out


def prroi_pool(features, rois, ph=7, pw=7, scale=1.0, out_nhwc=False):
    """features: NCHW-shaped tensor with ANY strides (dense NCHW or channels_last view);
    rois [R,5] device float32.  Returns NCHW-shaped [R,C,ph,pw] (channels_last strides when
    out_nhwc)."""
    _dev(features), _dev(rois)
    B, Cc, H, W_ = features.shape
    R = rois.shape[0]
    rois = rois.contiguous()
    if out_nhwc:
        buf = torch.zeros((R, ph, pw, Cc), device=features.device, dtype=torch.float32)
        out = buf.permute(0, 3, 1, 2)
    else:
        out = torch.zeros((R, Cc, ph, pw), device=features.device, dtype=torch.float32)
    fs, os_ = features.stride(), out.stride()
    check(lib().usot_prroi_pool_forward_f32(stream(), ptr(features), ptr(rois), ptr(out), R, Cc, H, W_, ph, pw,
                                            float(scale), fs[0], fs[1], fs[2], fs[3],
                                            os_[0], os_[1], os_[2], os_[3]), 'usot_prroi_pool_forward_f32')
    return out


def prroi_pool_backward(feature_shape, rois, top_diff, ph=7, pw=7, scale=1.0):
    """Feature gradient of Precise RoI Pooling (functional.py:71-73 -> prroi_pooling_gpu.c:46-77): dense NCHW
    [B,C,H,W] for top_diff [R,C,ph,pw]."""
    _dev(rois), _dev(top_diff)
    B, Cc, H, W_ = (int(v) for v in feature_shape)
    rois, top_diff = rois.contiguous(), top_diff.contiguous()
    out = torch.empty((B, Cc, H, W_), device=rois.device, dtype=torch.float32)        # zero-filled by the entry point
    check(lib().usot_prroi_pool_backward_f32(stream(), ptr(rois), ptr(top_diff), ptr(out), rois.shape[0], B, Cc, H, W_,
                                             ph, pw, float(scale)), 'usot_prroi_pool_backward_f32')
    return out


def prroi_pool_coor_backward(features, rois, top_data, top_diff, ph=7, pw=7, scale=1.0):
    """RoI gradient [R,5] (functional.py:74-76 -> prroi_pooling_gpu.c:79-113); column 0 (batch index) is zero."""
    _dev(features), _dev(rois)
    B, Cc, H, W_ = features.shape
    features, rois, top_data, top_diff = features.contiguous(), rois.contiguous(), top_data.contiguous(), top_diff.contiguous()
    out = torch.zeros((rois.shape[0], 5), device=rois.device, dtype=torch.float32)
    check(lib().usot_prroi_pool_coor_backward_f32(stream(), ptr(features), ptr(rois), ptr(top_data), ptr(top_diff), ptr(out),
                                                  rois.shape[0], B, Cc, H, W_, ph, pw, float(scale)),
          'usot_prroi_pool_coor_backward_f32')
    return out


def to_nhwc(t):
    """NCHW-shaped tensor (any strides) -> dense NHWC buffer [N,H,W,C] via the permute kernel."""
    _dev(t)
    N, Cc, H, W_ = t.shape
    s = t.stride()
    if s[1] == 1 and s[3] == Cc and s[2] == W_ * Cc and s[0] == H * W_ * Cc:
        return t.permute(0, 2, 3, 1)            # already channels-last in memory
    out = torch.empty((N, H, W_, Cc), device=t.device, dtype=torch.float32)
    check(lib().usot_permute4_f32(stream(), ptr(t), ptr(out), N, H, W_, Cc, s[0], s[2], s[3], s[1]),
          'usot_permute4_f32')
    return out


def to_nchw(t_nhwc):
    """Dense NHWC buffer -> dense NCHW tensor via the permute kernel."""
    _dev(t_nhwc)
    N, H, W_, Cc = t_nhwc.shape
    s = t_nhwc.stride()
    out = torch.empty((N, Cc, H, W_), device=t_nhwc.device, dtype=torch.float32)
    check(lib().usot_permute4_f32(stream(), ptr(t_nhwc), ptr(out), N, Cc, H, W_, s[0], s[3], s[1], s[2]),
          'usot_permute4_f32')
    return out


def decode(cls, cls_mem, bbox, window, S, instance_size, stride, ratio, penalty_k, window_influence, tw, th,
           out=None):
    _dev(cls), _dev(cls_mem), _dev(bbox), _dev(window, torch.float64)
    if out is None:
        out = torch.empty(8, device=cls.device, dtype=torch.float64)
    check(lib().usot_decode_f32(stream(), ptr(cls), ptr(cls_mem), ptr(bbox), ptr(window), ptr(out), S,
                                instance_size, stride, float(ratio), float(penalty_k),
                                float(window_influence), float(tw), float(th)), 'usot_decode_f32')
    return out


def conv2d_bf16(x, w, bias, *, KH, KW, stride=1, pad=(0, 0), dil=(1, 1), res=None, act=ACT_NONE, tile=0, out_f32=False,
                act2=ACT_NONE, act_split=0, groups=1):
    """x NHWC bf16|fp16 [N,H,W,Cin] (groups > 1: [G,N,H,W,Cin]), w packed same dtype [G*Cout, KH*KW*Cin],
    bias fp32 [G*Cout] -> y NHWC [N,OH,OW,Cout] ([G,N,OH,OW,Cout]) in the same dtype, or fp32 with out_f32."""
    lp = x.dtype
    if lp not in (torch.bfloat16, torch.float16):
        raise HipError('conv2d_bf16 takes bf16 or fp16 tensors')
    _dev(x, lp), _dev(w, lp)
    N, H, W_, Cin = x.shape[-4:]
    Cout = w.shape[0] // groups
    OH = (H + 2 * pad[0] - dil[0] * (KH - 1) - 1) // stride + 1
    OW = (W_ + 2 * pad[1] - dil[1] * (KW - 1) - 1) // stride + 1
    shape = (N, OH, OW, Cout) if groups == 1 else (groups, N, OH, OW, Cout)
    y = torch.empty(shape, device=x.device, dtype=torch.float32 if out_f32 else lp)
    d = conv_desc(x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                  N=N, H=H, W=W_, Cin=Cin, OH=OH, OW=OW, Cout=Cout, KH=KH, KW=KW, stride=stride, pad=pad, dil=dil,
                  res=res.data_ptr() if res is not None else None, act=act, act2=act2, act_split=act_split, tile=tile,
                  groups=groups, x_gs=N * H * W_ * Cin, w_gs=Cout * KH * KW * Cin, b_gs=Cout, y_gs=N * OH * OW * Cout)
    check(lib().usot_conv2d_lp(stream(), C.byref(d), 1 if lp == torch.float16 else 0, int(out_f32)), 'usot_conv2d_lp')
    return y


def pw_pair_pack(w, cm, co, cn, which):
    """Filter bank `w` ([CO,CM] for which = 0, [CN,CO] for which = 1) in the fragment order of the fused
    pointwise-pair kernel (usot_pw_pair_layout)."""
    n = w.numel() // 8
    row, k0 = (C.c_int32 * n)(), (C.c_int32 * n)()
    got = lib().usot_pw_pair_layout(int(cm), int(co), int(cn), int(which), row, k0)
    if got != n:
        raise HipError('usot_pw_pair_layout(%d, %d, %d, %d) = %d, expected %d chunks' % (cm, co, cn, which, got, n))
    r = torch.tensor(list(row), dtype=torch.long, device=w.device)
    k = torch.tensor(list(k0), dtype=torch.long, device=w.device)
    cols = k[:, None] + torch.arange(8, device=w.device)[None, :]
    return w[r[:, None], cols].contiguous().reshape(-1)


def pw_pair_f32_pack(w):
    """fp32 filter bank [rows, K] -> the fragment order of usot_pw_pair_f32: (column block, round, quad, row in block, 4 k)."""
    rows, k = w.shape
    return w.reshape(rows // 16, 16, k // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous().reshape(-1)


def pw_pair_s16_pack(w):
    """fp32 filter bank [rows, K] -> the split-fp16 bank of usot_pw_pair_f32s as one float32 tensor: rows x K floats of fragments
    (column block of 16 rows, 32-k step, hi | lo, quad, row in block, 8 halves) followed by the rows' factors 1 / (2^e x 8)."""
    rows, k = w.shape
    assert rows % 16 == 0 and k % 32 == 0
    w = w.float()
    amax = w.abs().amax(1)
    e = torch.where(amax > 0, torch.floor(torch.log2(1024.0 / amax.clamp_min(1e-30))), torch.zeros_like(amax))
    e = torch.where(amax * torch.exp2(e) >= 1024.0, e - 1, e).clamp(-100, 110)
    sw = torch.exp2(e)
    ws = w * sw[:, None]
    hi = ws.half()
    lo = (ws - hi.float()).half()
    frag = lambda t: t.reshape(rows // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4)          # (cb, step, quad, row, 8)
    packed = torch.stack([frag(hi), frag(lo)], 2).contiguous().reshape(-1)                      # (cb, step, hi | lo, quad, row, 8)
    return torch.cat([packed.view(torch.float32), (1.0 / (sw * SPLIT16_X_SCALE)).float()]).contiguous()


def pw_pair_f32_supported(cm, co, cn):
    return bool(lib().usot_pw_pair_f32_supported(int(cm), int(co), int(cn)))


def pw_pair_f32(t2, w3, b3, res, w1, b1, act2=ACT_RELU, sliced=True, split16=False, ovf=None):
    """fp32 NHWC: Y = relu(t2 . w3^T + b3 + res), T = act2(Y . w1^T + b1); w3 [CO,CM], w1 [CN,CO] in natural order
    (packed here).  Returns (Y, T)."""
    for t in (t2, w3, b3, res, w1, b1):
        _dev(t)
    CM, (CO, CN) = t2.shape[-1], (w3.shape[0], w1.shape[0])
    M = t2.numel() // CM
    y = torch.empty(tuple(t2.shape[:-1]) + (CO,), device=t2.device, dtype=torch.float32)
    t = torch.empty(tuple(t2.shape[:-1]) + (CN,), device=t2.device, dtype=torch.float32)
    w3p, w1p = (pw_pair_s16_pack(w3), pw_pair_s16_pack(w1)) if split16 else (pw_pair_f32_pack(w3), pw_pair_f32_pack(w1))
    ws = pw_pair_f32_ws(M, CM, CO, CN, t2.device) if sliced else None
    d = pw_pair_desc(t2.data_ptr(), w3p.data_ptr(), b3.data_ptr(), res.data_ptr(), y.data_ptr(), w1p.data_ptr(), b1.data_ptr(),
                     t.data_ptr(), M, CM, CO, CN, act2, ws.data_ptr() if ws is not None else None,
                     ovf=ovf.data_ptr() if ovf is not None else None)
    fn = lib().usot_pw_pair_f32s if split16 else lib().usot_pw_pair_f32
    check(fn(stream(), C.byref(d)), 'usot_pw_pair_f32')
    if ws is not None:
        check(fn(stream(), C.byref(d)), 'usot_pw_pair_f32')      # a second launch finds the tickets reset
    return y, t


def pw_single_f32(x, w, b, res=None, act=ACT_NONE):
    """fp32 NHWC pointwise convolution on the small-M streaming kernel: y = act(x . w^T + b (+ res)); w [N,K] natural order."""
    _dev(x), _dev(w), _dev(b)
    K, N = x.shape[-1], w.shape[0]
    M = x.numel() // K
    y = torch.empty(tuple(x.shape[:-1]) + (N,), device=x.device, dtype=torch.float32)
    wp = pw_pair_f32_pack(w)
    check(lib().usot_pw_single_f32(stream(), ptr(x), ptr(wp), ptr(b), ptr(res) if res is not None else None, ptr(y), M, K, N, act),
          'usot_pw_single_f32')
    return y


def stream_conv3x3_f32(x, w, b, pad, dil, res=None, act=ACT_NONE):
    """fp32 NHWC 3x3 / stride-1 convolution on the small-M streaming kernel; w packed [N, 9*Cin] in (kh, kw, ci) order."""
    _dev(x), _dev(w), _dev(b)
    Nb, H, W_, Cin = x.shape
    N = w.shape[0]
    OH, OW = H + 2 * pad[0] - 2 * dil[0], W_ + 2 * pad[1] - 2 * dil[1]
    y = torch.empty((Nb, OH, OW, N), device=x.device, dtype=torch.float32)
    wp = pw_pair_f32_pack(w)
    check(lib().usot_stream_conv3x3_f32(stream(), ptr(x), ptr(wp), ptr(b), ptr(res) if res is not None else None, ptr(y),
                                        Nb, H, W_, Cin, OH, OW, N, pad[0], pad[1], dil[0], dil[1], act), 'usot_stream_conv3x3_f32')
    return y


def pw_triple_f32(x, w2, b2, w3, b3, res, w1, b1, pad=(1, 1), dil=(1, 1), act2=ACT_RELU):
    """fp32 NHWC: T2 = relu(conv3x3(x, w2) + b2); Y = relu(T2 . w3^T + b3 + res); T = act2(Y . w1^T + b1) in one launch.
    w2 packed [CM, 9*Cin] ((kh, kw, ci) order), w3 [CO, CM], w1 [CN, CO].  Returns (Y, T) as [Nb, OH, OW, C]."""
    for t in (x, w2, b2, w3, b3, res, w1, b1):
        _dev(t)
    Nb, H, W_, Cin = x.shape
    CM, CO, CN = w2.shape[0], w3.shape[0], w1.shape[0]
    OH, OW = H + 2 * pad[0] - 2 * dil[0], W_ + 2 * pad[1] - 2 * dil[1]
    y = torch.empty((Nb, OH, OW, CO), device=x.device, dtype=torch.float32)
    t = torch.empty((Nb, OH, OW, CN), device=x.device, dtype=torch.float32)
    w2p, w3p, w1p = pw_pair_f32_pack(w2), pw_pair_f32_pack(w3), pw_pair_f32_pack(w1)
    d = pw_pair_desc(None, w3p.data_ptr(), b3.data_ptr(), res.data_ptr(), y.data_ptr(), w1p.data_ptr(), b1.data_ptr(), t.data_ptr(),
                     Nb * OH * OW, CM, CO, CN, act2)
    check(lib().usot_pw_triple_f32(stream(), ptr(x), ptr(w2p), ptr(b2), C.byref(d), Nb, H, W_, Cin, OH, OW, pad[0], pad[1], dil[0], dil[1]),
          'usot_pw_triple_f32')
    return y, t


def pw_pair_supported(cm, co, cn):
    return bool(lib().usot_pw_pair_supported(int(cm), int(co), int(cn)))


def pw_pair_desc(t2, w3p, b3, res, y, w1, b1, t, M, CM, CO, CN, act2, ws=None, t2_parts=0, t2_bias=None, res_parts=0, res_bias=None,
                 ovf=None):
    d = PwPairDesc()
    d.t2, d.w3p, d.res, d.w1, d.b3, d.b1, d.y, d.t = t2, w3p, res, w1, b3, b1, y, t
    d.M, d.CM, d.CO, d.CN, d.act2 = M, CM, CO, CN, act2
    d.ws = ws
    d.t2_parts, d.t2_bias = t2_parts, t2_bias
    d.res_parts, d.res_bias = res_parts, res_bias
    d.ovf = ovf or None
    return d


def bneck_desc(x, w1, b1, w2, b2, w3c, b3c, wn, bn, y, t, N, H, W):
    d = BneckDesc()
    d.x, d.w1, d.w2, d.w3c, d.wn, d.b1, d.b2, d.b3c, d.bn, d.y, d.t = x, w1, w2, w3c, wn, b1, b2, b3c, bn, y, t
    d.N, d.H, d.W = N, H, W
    return d


def pw_pair_f32_ws(M, CM, CO, CN, device):
    """Zero-initialised workspace of the channel-sliced form of usot_pw_pair_f32, or None when the shape runs unsliced."""
    n = lib().usot_pw_pair_f32_ws_floats(int(M), int(CM), int(CO), int(CN))
    return torch.zeros(n, device=device, dtype=torch.float32) if n > 0 else None


def pw_pair(t2, w3, b3, res, w1, b1, act2=ACT_RELU):
    """Fused conv3 + residual + ReLU -> next 1x1 conv (+ act2) on low-precision [M, C] maps.
    t2 [M,CM], w3 [CO,CM] (natural rows), b3 fp32 [CO], res [M,CO], w1 [CN,CO], b1 fp32 [CN] -> (y [M,CO], t [M,CN])."""
    lp = t2.dtype
    if lp not in (torch.bfloat16, torch.float16):
        raise HipError('pw_pair takes bf16 or fp16 tensors')
    for a in (t2, w3, res, w1):
        _dev(a, lp)
    _dev(b3), _dev(b1)
    M, CM = t2.shape
    CO, CN = w3.shape[0], w1.shape[0]
    w3p, w1 = pw_pair_pack(w3, CM, CO, CN, 0), pw_pair_pack(w1, CM, CO, CN, 1)
    y = torch.empty((M, CO), device=t2.device, dtype=lp)
    t = torch.empty((M, CN), device=t2.device, dtype=lp)
    d = pw_pair_desc(t2.data_ptr(), w3p.data_ptr(), b3.data_ptr(), res.data_ptr(), y.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                     t.data_ptr(), M, CM, CO, CN, act2)
    check(lib().usot_pw_pair_lp(stream(), C.byref(d), 1 if lp == torch.float16 else 0), 'usot_pw_pair_lp')
    return y, t


def crop_resize(frame_u8, out, x0, y0, win, fill):
    """frame_u8: device uint8 [H,W,3]; out: device float32 [3,S,S] (written in place)."""
    _dev(frame_u8, torch.uint8), _dev(out)
    if not frame_u8.is_contiguous() or not out.is_contiguous():
        raise HipError('crop_resize needs a dense HWC uint8 frame and a dense CHW output')
    H, W_, _ = frame_u8.shape
    S = out.shape[-1]
    check(lib().usot_crop_resize_u8_f32(stream(), ptr(frame_u8), ptr(out), H, W_, int(x0), int(y0), int(win), S,
                                        int(fill[0]), int(fill[1]), int(fill[2])), 'usot_crop_resize_u8_f32')
    return out
