"""ctypes binding of libgcc_hip.so (include/gcc_hip.h).  There is NO fallback: if the library is
missing or a call fails, this raises -- the product never routes through PyTorch eager or the
CPU oracle."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GCC_HIP_LIB') or os.path.join(_HERE, 'libgcc_hip.so')     # GCC_HIP_LIB: another build of the same ABI (A/B runs)

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH = 0, 1, 2, 3
GCC_HIP_ABI = 603
WGRAD_GROUP_MAX = 32     # include/gcc_hip.h GCC_WGRAD_GROUP_MAX
CHANSUM_GROUP_MAX = 24   # include/gcc_hip.h GCC_CHANSUM_GROUP_MAX
SPECTRAL_GROUP_MAX = 8    # include/gcc_hip.h GCC_SPECTRAL_GROUP_MAX
CHANSUM_SMALL_MAX_PIXELS = 16384     # include/gcc_hip.h GCC_HIP_ABI: the generation of struct layouts / option ids these bindings were written for


class GccError(RuntimeError):
    pass


class conv_plan_t(C.Structure):
    """include/gcc_hip.h gcc_conv_plan_t: the tile plan of a convolution call (every field 0 = the library's default)"""
    _fields_ = [(n, C.c_int) for n in ('tile_families', 'big_min', 'big_nk', 'pair', 'halo_hc', 'wgrad_wgs_big', 'wgrad_wgs')]


PLAN_FIELDS = tuple(n for n, _ in conv_plan_t._fields_)


class conv_t(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('N', 'H', 'W', 'Ci', 'Co', 'KH', 'KW', 'stride', 'pad',
                                       'ldx', 'xoff', 'ldy', 'yoff')] + [('plan', conv_plan_t)]


class wgrad_item_t(C.Structure):
    """include/gcc_hip.h gcc_wgrad_item_t: one entry of a grouped weight gradient"""
    _fields_ = [('c', conv_t), ('x', C.c_void_p), ('dy', C.c_void_p), ('dw', C.c_void_p), ('accumulate', C.c_int)]


class chansum_item_t(C.Structure):
    """include/gcc_hip.h gcc_chansum_item_t"""
    _fields_ = [('x', C.c_void_p), ('ld', C.c_int), ('off', C.c_int), ('C', C.c_int), ('pixels', C.c_size_t), ('out', C.c_void_p),
                ('accumulate', C.c_int)]


class sn_item_t(C.Structure):
    """include/gcc_hip.h gcc_sn_item_t: one layer of a grouped spectral-norm power iteration"""
    _fields_ = [('w_bar', C.c_void_p), ('u', C.c_void_p), ('v', C.c_void_p), ('R', C.c_int), ('C', C.c_int), ('T', C.c_int),
                ('t_out', C.c_void_p), ('sigma_out', C.c_void_p), ('w', C.c_void_p), ('wt', C.c_void_p)]


class epilogue_t(C.Structure):
    _fields_ = [('bias', C.c_void_p), ('act', C.c_int), ('slope', C.c_float), ('stats_partial', C.c_void_p),
                ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t), ('bn', C.c_void_p),
                ('y2', C.c_void_p), ('ldy2', C.c_int), ('y2off', C.c_int), ('y2_mode', C.c_int), ('y2_gate', C.c_void_p)]


class bnact_t(C.Structure):
    _fields_ = [('scale', C.c_void_p), ('shift', C.c_void_p), ('gate', C.c_void_p), ('gate_after_act', C.c_int),
                ('act', C.c_int), ('slope', C.c_float), ('act2', C.c_int), ('drop_p', C.c_float),
                ('seed', C.c_uint64), ('groups', C.c_int), ('ld_residual', C.c_int), ('residual', C.c_void_p)]


class bn_t(C.Structure):
    _fields_ = [('gamma', C.c_void_p), ('beta', C.c_void_p), ('eps', C.c_float), ('momentum', C.c_float), ('count', C.c_double),
                ('running_mean', C.c_void_p), ('running_var', C.c_void_p), ('mean', C.c_void_p), ('rstd', C.c_void_p),
                ('scale', C.c_void_p), ('shift', C.c_void_p), ('tail_ws', C.c_void_p), ('tail_ws_bytes', C.c_size_t),
                ('finalize_in_launch', C.c_int), ('pad_', C.c_int)]


class bnact_bwd_t(C.Structure):
    _fields_ = [('bn', C.c_int), ('bn_eval', C.c_int), ('mean', C.c_void_p), ('rstd', C.c_void_p),
                ('gamma', C.c_void_p), ('beta', C.c_void_p), ('gate', C.c_void_p), ('gate_after_act', C.c_int),
                ('act', C.c_int), ('slope', C.c_float), ('act2', C.c_int), ('drop_p', C.c_float),
                ('seed', C.c_uint64), ('dgamma', C.c_void_p), ('dbeta', C.c_void_p), ('dalpha', C.c_void_p),
                ('groups', C.c_int), ('flags', C.c_int)]


class adam_tensor_t(C.Structure):
    _fields_ = [('p', C.c_void_p), ('g', C.c_void_p), ('m', C.c_void_p), ('v', C.c_void_p),
                ('numel', C.c_int64), ('l1', C.c_float), ('grad_scale', C.c_float)]


class pack_desc_t(C.Structure):
    _fields_ = [('master', C.c_void_p), ('w', C.c_void_p), ('wt', C.c_void_p), ('rows', C.c_int), ('taps', C.c_int),
                ('cols', C.c_int), ('colsp', C.c_int), ('rowsp', C.c_int), ('row_split', C.c_int),
                ('col_split', C.c_int), ('pad_', C.c_int)]


class pack_item_t(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('tensor', 'kind', 'a', 'b', 'c', 'pad_')]


class adam_chunk_t(C.Structure):
    _fields_ = [('tensor', C.c_int), ('pad_', C.c_int), ('offset', C.c_int64)]


# gcc_set_option ids (enum in include/gcc_hip.h)
(OPT_IGEMM_GLDS, OPT_IGEMM_HEAD, OPT_IGEMM_THIN, OPT_WGRAD_BIG, OPT_BN_SWEEPS, OPT_BN_MAXBLK, OPT_BN_REDUCE_THREADS,
 OPT_BN_REDUCE_CAP, OPT_INORM_LPP, OPT_IGEMM_FORCE_BC, OPT_IGEMM_FORCE_KSPLIT, OPT_IGEMM_NARROW, OPT_WGRAD_BIG_MIN_TILES,
 OPT_FUSE_BN, OPT_BN_BWD_SMALL, OPT_WGRAD_ROW_TABLE, OPT_IGEMM_HALO, OPT_FUSE_BN_PARTIAL_KB, OPT_INORM_GRID, OPT_IGEMM_STAGES,
 OPT_WGRAD_TS, OPT_HALO_XCD_COLS) = range(22)
OPT_COUNT = 22
OPT_NAMES = ('IGEMM_GLDS', 'IGEMM_HEAD', 'IGEMM_THIN', 'WGRAD_BIG', 'BN_SWEEPS', 'BN_MAXBLK', 'BN_REDUCE_THREADS', 'BN_REDUCE_CAP',
             'INORM_LPP', 'IGEMM_FORCE_BC', 'IGEMM_FORCE_KSPLIT', 'IGEMM_NARROW', 'WGRAD_BIG_MIN_TILES', 'FUSE_BN', 'BN_BWD_SMALL',
             'WGRAD_ROW_TABLE', 'IGEMM_HALO', 'FUSE_BN_PARTIAL_KB', 'INORM_GRID', 'IGEMM_STAGES', 'WGRAD_TS', 'HALO_XCD_COLS')

_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_Z = C.c_size_t

# name -> (restype, argtypes): every symbol include/gcc_hip.h declares
PROTOTYPES = {
    'gcc_strerror': (C.c_char_p, [_I]),
    'gcc_version': (_I, []),
    'gcc_launch_count': (C.c_longlong, [_I]),
    'gcc_device_error': (_I, [_I]),
    'gcc_set_option': (_I, [_I, _I]),
    'gcc_get_option': (_I, [_I]),
    'gcc_options_default': (_I, []),
    'gcc_conv_tile': (_I, [C.POINTER(conv_t), _I]),
    'gcc_conv_stat_tiles': (_I, [C.POINTER(conv_t), _I]),
    'gcc_conv_route': (_I, [C.POINTER(conv_t), _I, C.POINTER(epilogue_t)]),
    'gcc_conv_y2_supported': (_I, [C.POINTER(conv_t), _I, C.POINTER(epilogue_t)]),
    'gcc_conv_workspace': (_Z, [C.POINTER(conv_t), _I]),
    'gcc_conv_fprop': (_I, [C.POINTER(conv_t), _P, _P, _P, C.POINTER(epilogue_t), _P]),
    'gcc_conv_dgrad': (_I, [C.POINTER(conv_t), _P, _P, _P, C.POINTER(epilogue_t), _P]),
    'gcc_conv_wgrad_workspace': (_Z, [C.POINTER(conv_t)]),
    'gcc_conv_wgrad': (_I, [C.POINTER(conv_t), _P, _P, _P, _I, _P, _Z, _P]),
    'gcc_conv_wgrad_seg': (_I, [C.POINTER(conv_t), _P, _P, _P, _I, _I, _I, _I, _I, _P, _Z, _P]),
    'gcc_conv_wgrad_group_table_bytes': (_Z, []),
    'gcc_conv_wgrad_group_workspace': (_Z, [C.POINTER(wgrad_item_t), _I]),
    'gcc_conv_wgrad_group_prepare': (_I, [C.POINTER(wgrad_item_t), _I, _P, _Z, _P]),
    'gcc_conv_wgrad_group_run': (_I, [_P, _P, _P]),
    'gcc_pack_weights': (_I, [_P, _I, _I, _I, _P, _P, _P]),
    'gcc_pack_weights_multi': (_I, [_P, _P, _I, _P]),
    'gcc_nchw_f32_to_nhwc_bf16': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'gcc_nhwc_bf16_to_nchw_f32': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'gcc_nhwc_copy': (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _Z, _P]),
    'gcc_nhwc_pack_pair': (_I, [_P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _Z, _P]),
    'gcc_nhwc_add': (_I, [_P, _I, _I, _P, _I, _I, _I, _Z, _P]),
    'gcc_bn_finalize': (_I, [_P, _I, _I, C.c_double, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P]),
    'gcc_in_finalize': (_I, [_P, _I, _I, _I, C.c_double, _F, _P, _P, _P, _P, _P]),
    'gcc_write_i32': (_I, [_P, _P, _I, _P]),
    'gcc_image_pool_query': (_I, [_P, _P, _P, _P, _I, _Z, _I, _P]),
    'gcc_replay_begin': (_I, [_P]),
    'gcc_replay_end': (_I, [_P, _I]),
    'gcc_replay_run': (_I, [_P]),
    'gcc_replay_tag_next': (_I, [_I]),
    'gcc_replay_patch': (_I, [_P, _I, _I, _P, _Z]),
    'gcc_replay_info': (C.c_longlong, [_P, _I]),
    'gcc_replay_destroy': (_I, [_P]),
    'gcc_event_create': (_I, [_P]),
    'gcc_event_destroy': (_I, [_P]),
    'gcc_event_record': (_I, [_P, _P]),
    'gcc_stream_wait_event': (_I, [_P, _P]),
    'gcc_adam_factors': (_I, [_F, _F, _I, _P]),
    'gcc_inorm_fwd': (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P, _Z, _P]),
    'gcc_inorm_bwd': (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _F, _P, _P, _P, _Z, _P]),
    'gcc_bn_bwd_one_launch': (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _Z, _I, _F, _P, _P, _P, _P, _P, _P, _Z, _P]),
    'gcc_bn_bwd_one_launch_ex': (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _Z, _I, _I, _F, _F, C.c_uint64, _P, _P, _P, _P, _P, _P, _P,
                                      _P, _Z, _P]),
    'gcc_channel_stats_tiles': (_I, [_Z, _I]),
    'gcc_channel_stats': (_I, [_P, _I, _I, _I, _Z, _I, _P, _P]),
    'gcc_reflect_pad': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'gcc_dwconv3x3_reflect': (_I, [_I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P]),
    'gcc_dwconv3x3_wgrad_workspace': (_Z, [_I, _I, _I, _I]),
    'gcc_dwconv3x3_reflect_wgrad': (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P, _Z, _P]),
    'gcc_bn_eval_coeffs': (_I, [_P, _P, _P, _P, _F, _I, _P, _P, _P]),
    'gcc_bnact_fwd': (_I, [C.POINTER(bnact_t), _P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _Z, _P]),
    'gcc_conv_bn_act_workspace': (_Z, [C.POINTER(conv_t), _I]),
    'gcc_conv_bn_act': (_I, [C.POINTER(conv_t), _I, _P, _P, _P, C.POINTER(bn_t), C.POINTER(bnact_t), _P, _I, _I, _P, _I, _I, _P, _Z, _P]),
    'gcc_bnact_bwd_workspace': (_Z, [_I, _Z]),
    'gcc_bnact_bwd': (_I, [C.POINTER(bnact_bwd_t), _P, _I, _I, _P, _I, _I, _P, _I, _I, _P, _I, _I, _P, _I, _I,
                           _I, _Z, _P, _Z, _P]),
    'gcc_bnact_bwd_ex': (_I, [C.POINTER(bnact_bwd_t), _I, _F, _P, _I, _I, _P, _I, _I, _P, _I, _I, _P, _I, _I, _P,
                              _I, _I, _I, _Z, _P, _Z, _P]),
    'gcc_channel_sum': (_I, [_P, _I, _I, _I, _Z, _P, _I, _P, _Z, _P]),
    'gcc_channel_sum_workspace': (_Z, [_I, _Z]),
    'gcc_channel_sum_group': (_I, [C.POINTER(chansum_item_t), _I, _P]),
    'gcc_gate_mask': (_I, [_P, _F, _P, _I, _P]),
    'gcc_gan_loss': (_I, [_I, _I, _I, _P, _I, _I, _Z, _F, _P, _I, _P, _P, _Z, _P]),
    'gcc_gan_loss_ex': (_I, [_I, _I, _I, _P, _I, _I, _Z, _P, _P, _F, _P, _I, _P]),
    'gcc_arch_coeffs': (_I, [_P, _P, _P, _P, _F, _P, _P, _P, _P]),
    'gcc_spectral_workspace': (_Z, [_I, _I, _I]),
    'gcc_spectral_power_iteration': (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _Z, _P]),
    'gcc_spectral_power_iteration_pack': (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _Z, _P]),
    'gcc_spectral_group_workspace': (_Z, [C.POINTER(sn_item_t), _I]),
    'gcc_spectral_power_iteration_pack_group': (_I, [C.POINTER(sn_item_t), _I, _P, _Z, _P]),
    'gcc_spectral_grad': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _Z, _P]),
    'gcc_resample_u8': (_I, [_P, _I, _I, _Z, _P, _I, _I, _P, _P, _I, _P, _P, _I, _P, _P]),
    'gcc_crop_flip_normalize': (_I, [_P, _I, _I, _Z, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    'gcc_crop_convert': (_I, [_P, _I, _I, _Z, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P]),
    'gcc_argmax_channels': (_I, [_P, _I, _I, _Z, _P, _P]),
    'gcc_confusion_hist': (_I, [_P, _P, _Z, _I, _P, _P]),
    'gcc_psnr_workspace': (_Z, []),
    'gcc_psnr_y_sse': (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _Z, _P]),
    'gcc_ssim_y_sum': (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _Z, _P]),
    'gcc_activation_stats_workspace': (_Z, [_I, _I]),
    'gcc_activation_stats': (_I, [_P, _I, _I, _I, _P, _P, _P, _Z, _P]),
    'gcc_frechet_workspace': (_Z, [_I]),
    'gcc_frechet_distance': (_I, [_P, _P, _P, _P, _I, _I, C.c_double, _P, _P, _Z, _P]),
    'gcc_attention_fwd': (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P]),
    'gcc_attention_bwd': (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P]),
    'gcc_l1_loss': (_I, [_P, _I, _I, _P, _I, _I, _I, _Z, _F, _P, _I, _P, _I, _I, _P, _Z, _P]),
    'gcc_mse_loss': (_I, [_P, _I, _I, _P, _I, _I, _I, _Z, _F, _P, _I, _P, _I, _I, _P, _Z, _P]),
    'gcc_loss_workspace': (_Z, [_Z, _I]),
    'gcc_prelu': (_I, [_I, _P, _I, _P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _Z, _P]),
    'gcc_maxpool2x2': (_I, [_I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    'gcc_pool_linear_fwd': (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P]),
    'gcc_pool_linear_bwd': (_I, [_P, _I, _P, _P, _I, _I, _I, _P, _I, _P, _P, _P]),
    'gcc_distill_workspace': (_Z, [_I, _I, _I]),
    'gcc_distill_fwd': (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P, _Z, _P]),
    'gcc_distill_bwd': (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _F, _F, _P, _I, _I, _P, _Z, _P]),
    'gcc_adam_step': (_I, [_P, _P, _I, _I, _F, _F, _F, _F, _I, _P]),
    'gcc_fill_f32': (_I, [_P, _F, _Z, _P]),
    'gcc_add_f32': (_I, [_P, _P, _Z, _P]),
    'gcc_clamp_f32': (_I, [_P, _F, _F, _Z, _P]),
    'gcc_scalar_op': (_I, [_I, _P, _P, _P, _F, _F, _P, _P]),
    'gcc_comm_unique_id': (_I, [_P]),
    'gcc_comm_init': (_I, [C.POINTER(C.c_void_p), _I, _I, _P]),
    'gcc_comm_allreduce_sum_f32': (_I, [_P, _P, _Z, _P]),
    'gcc_comm_allreduce_sum_bf16': (_I, [_P, _P, _Z, _P]),
    'gcc_cast_f32_bf16': (_I, [_P, _P, _Z, _P]),
    'gcc_cast_bf16_f32': (_I, [_P, _P, _Z, _P]),
    'gcc_comm_rank': (_I, [_P]),
    'gcc_comm_world': (_I, [_P]),
    'gcc_comm_count': (_I, [_P]),
    'gcc_comm_destroy': (_I, [_P]),
    'gcc_comm_last_error': (C.c_char_p, []),
}

_lib = None


def load():
    """Load the in-tree library; raises GccError when it has not been built (run
    ``python -c 'import __graft_entry__ as g; g.build()'`` or gcc_amd/csrc/build.sh)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GccError('libgcc_hip.so is not built (%s missing): the HIP kernels are the only '
                       'implementation of this path -- build them with gcc_amd/csrc/build.sh' % LIB_PATH)
    # PyTorch-ROCm ships its own HIP runtime: it must be resident first so that this library binds to
    # the same libamdhip64 (and the same device/stream state) instead of initialising a second one
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.gcc_version()
    if got != GCC_HIP_ABI:
        raise GccError('%s was built from another generation of include/gcc_hip.h (gcc_version() = %d, these bindings are '
                       'for GCC_HIP_ABI %d): struct layouts and option ids differ -- rebuild with gcc_amd/csrc/build.sh'
                       % (LIB_PATH, got, GCC_HIP_ABI))
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().gcc_strerror(rc).decode()
        raise GccError('%s failed: %s (%d)' % (what or 'gcc call', msg, rc))
