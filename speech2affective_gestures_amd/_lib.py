"""ctypes binding of include/s2ag_hip.h.  There is NO fallback: if libs2ag_hip.so is missing or a
symbol is absent, importing the compute path raises."""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, 'libs2ag_hip.so')

vp, ci, cf, cu, cll = C.c_void_p, C.c_int, C.c_float, C.c_uint, C.c_longlong


class ConvGeom(C.Structure):
    _fields_ = [(n, ci) for n in ('N', 'Lin', 'Lout', 'Cin', 'Cout', 'ksize', 'stride', 'pad', 'dil', 'ldx', 'ldy',
                                   'w_tap_major')]


class Epilogue(C.Structure):
    _fields_ = [('act', ci), ('slope', cf), ('drop_p', cf), ('rng', vp), ('site', cu)]


ACT_NONE, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2
PG, PE = C.POINTER(ConvGeom), C.POINTER(Epilogue)

# name -> argtypes (every function returns int); mirrors include/s2ag_hip.h one to one
class SpmvJob(C.Structure):
    _fields_ = [('rowptr', C.c_void_p), ('col', C.c_void_p), ('val', C.c_void_p), ('x', C.c_void_p), ('y', C.c_void_p),
                ('nrows', C.c_int)]


class WnJob(C.Structure):
    _fields_ = [('v', C.c_void_p), ('g', C.c_void_p), ('w', C.c_void_p), ('norm', C.c_void_p), ('dw', C.c_void_p),
                ('dv', C.c_void_p), ('dg', C.c_void_p), ('rows', C.c_int), ('cols', C.c_int), ('ksize', C.c_int)]


class WgradJob(C.Structure):     # s2ag_wgrad_job
    _fields_ = [('gy', C.c_void_p), ('x', C.c_void_p), ('dw', C.c_void_p), ('dbias', C.c_void_p), ('geom', ConvGeom)]


class BF16Conv(C.Structure):      # s2ag_bf16_conv_args
    _fields_ = [('x', vp), ('w', vp), ('bias', vp), ('y', vp), ('N', ci), ('Lq', ci), ('Lin', ci), ('x_clip', cll),
                ('ldx', ci), ('pos_mul', ci), ('pos_off', ci), ('pos_tap', ci), ('ks', ci), ('Cp', ci), ('Cvalid', ci),
                ('Cout', ci), ('CoutS', ci), ('y_clip', cll), ('y_row', ci), ('y_off', ci), ('out_f32', ci),
                ('phases', ci), ('w_phase', cll), ('y_phase', ci), ('q_total', ci), ('mask_cols', ci), ('post_y', vp),
                ('post_act', ci), ('post_cols', ci), ('post_slope', cf), ('post_drop', cf), ('post_rng', vp),
                ('post_site', cu)]


class BF16Wgrad(C.Structure):     # s2ag_bf16_wgrad_args
    _fields_ = [('gy', vp), ('x', vp), ('dw', vp), ('db', vp), ('N', ci), ('Lq', ci), ('Lin', ci), ('x_clip', cll),
                ('ldx', ci), ('ldg', ci), ('pos_mul', ci), ('pos_off', ci), ('pos_tap', ci), ('ks', ci), ('Cp', ci),
                ('Cvalid', ci), ('Cout', ci), ('Cin', ci), ('d_co', cll), ('d_t', ci), ('d_c', ci), ('flat_cin', ci),
                ('ks_out', ci)]


class BnFoldArgs(C.Structure):    # s2ag_bn_fold_args
    _fields_ = [('ticket', vp), ('gamma', vp), ('beta', vp), ('running_mean', vp), ('running_var', vp),
                ('num_batches_tracked', vp), ('eps', cf), ('momentum', cf), ('repeat', ci), ('scale', vp), ('shift', vp),
                ('mean', vp), ('invstd', vp)]


class Wave12Bwd(C.Structure):     # s2ag_wave12_bwd_args
    _fields_ = [('x', vp), ('packed', vp), ('b1', vp), ('scale1', vp), ('shift1', vp), ('mean1', vp), ('invstd1', vp),
                ('gamma1', vp), ('slope', cf), ('dz', vp), ('dz_f32', ci), ('z2', vp), ('ca2', vp), ('cb2', vp), ('cc2', vp),
                ('part_w2', vp), ('part_s', vp), ('stats', vp), ('ticket', vp),
                ('dgamma1', vp), ('dbeta1', vp), ('ca1', vp), ('cb1', vp), ('cc1', vp), ('dw2', vp), ('dw1', vp),
                ('N', ci), ('Lin', ci), ('L1', ci), ('L2', ci), ('pad', ci)]


class BF16PackJob(C.Structure):   # s2ag_bf16_pack_job
    _fields_ = [('src', vp), ('dst', vp), ('rows', ci), ('taps', ci), ('Cp', ci), ('cols', ci), ('tap0', ci),
                ('tap_step', ci), ('src_taps', ci), ('s_o', cll), ('s_t', ci), ('s_c', ci), ('flat_cin', ci)]


class BF16Tcn(C.Structure):       # s2ag_bf16_tcn_args
    _fields_ = [('x', vp), ('h1', vp * 4), ('sign', vp * 4), ('y', vp * 4), ('wfrag', vp), ('bias', vp * 8), ('dil', ci * 4),
                ('n_blocks', ci), ('n_clips', ci), ('T', ci), ('C', ci), ('drop_p', cf), ('rng', vp), ('site', cu * 8),
                ('gy', vp), ('gx', vp), ('gp1', vp * 4), ('gp2', vp * 4), ('keep', vp)]


class Tcn32(C.Structure):         # s2ag_tcn32_args
    _fields_ = [('x', vp), ('h1', vp * 4), ('h2', vp * 4), ('y', vp * 4), ('wfrag', vp), ('bias', vp * 8), ('dil', ci * 4),
                ('n_blocks', ci), ('n_clips', ci), ('T', ci), ('C', ci), ('drop_p', cf), ('rng', vp), ('site', cu * 8),
                ('keep', vp), ('gy', vp), ('gx', vp), ('gp1', vp * 4), ('gp2', vp * 4)]


MAX_JOBS = 8
MAX_WGRAD_JOBS = 4
BF16_MAX_PACK = 32
BF16_MAX_WGRAD_JOBS = 8
TCN_MAX_BLOCKS = 4

SIGNATURES = {
    's2ag_abi_version': [],
    's2ag_set_option': [C.c_char_p, ci],
    's2ag_get_option': [C.c_char_p],
    's2ag_det_flavour': [],
    's2ag_set_deterministic': [vp, vp],
    's2ag_conv1d_nlc_fwd': [vp, vp, vp, vp, PG, PE, vp],
    's2ag_conv1d_nlc_bwd_data': [vp, vp, vp, PG, ci, vp],
    's2ag_conv1d_nlc_bwd_weight': [vp, vp, vp, vp, PG, ci, vp],
    's2ag_conv1d_nlc_bwd_pair': [vp, vp, vp, vp, vp, vp, PG, vp],
    's2ag_conv1d_nlc_bwd_weight_multi': [vp, ci, vp],
    's2ag_colsum': [vp, ci, ci, ci, vp, vp, ci, vp],
    's2ag_colstats_f64': [vp, ci, ci, ci, vp, vp, vp],
    's2ag_bn_coeffs': [vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, cf, cf, ci, vp, vp, vp, vp, vp],
    's2ag_bn_partial_rows': [ci, ci, ci],
    's2ag_bn_fwd_stats': [vp, ci, ci, ci, vp, ci, vp, vp, vp, vp, vp, cf, cf, ci, vp, vp, vp, vp, vp, vp, vp],
    's2ag_bn_bwd_stats': [vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, cf, vp, ci, vp, vp, ci, vp, vp, vp, vp, vp],
    's2ag_conv_stats_rows': [PG],
    's2ag_conv1d_nlc_fwd_stats': [vp, vp, vp, vp, PG, PE, vp, vp, vp],
    's2ag_bn_fold': [vp, ci, ci, ci, vp, ci, vp, vp, vp, vp, vp, cf, cf, ci, vp, vp, vp, vp, vp],
    's2ag_bn_fold_apply_supported': [ci, ci, ci],
    's2ag_bn_fold_apply': [vp, ci, ci, ci, vp, ci, vp, vp, vp, vp, vp, cf, cf, ci, vp, vp, vp, vp, vp, ci, cf, vp, ci, vp],
    's2ag_bn_apply': [vp, ci, ci, ci, vp, vp, cf, vp, ci, vp],
    's2ag_bn_fused_supported': [ci, ci],
    's2ag_bn_fused_partial_rows': [ci, ci, ci],
    's2ag_bn_set_error_flag': [vp],
    's2ag_bn_fwd_fused': [vp, ci, ci, ci, vp, ci, vp, vp, vp, vp, vp, cf, cf, ci, vp, vp, vp, vp, vp, vp, cf, vp, ci, vp],
    's2ag_bn_bwd_fused': [vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, cf, vp, ci, vp, vp, ci, vp, vp, vp, vp, vp, ci, vp],
    's2ag_bn_bwd_reduce': [vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, cf, vp, vp, vp],
    's2ag_bn_bwd_coeffs': [vp, vp, vp, ci, ci, ci, vp, vp, ci, vp, vp, vp],
    's2ag_bn_bwd_apply': [vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, cf, vp, vp, vp, ci, vp],
    's2ag_add_act': [vp, ci, vp, ci, vp, ci, ci, ci, cf, vp],
    's2ag_epilogue_bwd': [vp, ci, vp, ci, vp, ci, ci, ci, PE, vp],
    's2ag_embedding_fwd': [vp, vp, ci, ci, ci, vp, ci, PE, vp],
    's2ag_embedding_bwd': [vp, vp, ci, ci, ci, ci, vp, ci, PE, vp],
    's2ag_weight_norm_fwd': [vp, vp, ci, ci, ci, vp, vp, vp],
    's2ag_weight_norm_bwd': [vp, vp, vp, vp, ci, ci, ci, vp, vp, vp],
    's2ag_spmv': [vp, vp, vp, vp, vp, ci, ci, vp],
    's2ag_timestamp': [vp, vp],
    's2ag_install_crash_handler': [ci],
    's2ag_spmv_multi': [vp, ci, ci, vp],
    's2ag_weight_norm_multi': [vp, ci, ci, vp],
    's2ag_transpose': [vp, ci, ci, vp, vp],
    's2ag_gru_seq_needs_transposed': [ci],
    's2ag_gru_seq_fwd': [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, PE, vp],
    's2ag_gru_seq_bwd': [vp, ci, ci, vp, vp, vp, vp, vp, ci, ci, ci, PE, vp],
    's2ag_split_k_padded': [ci],
    's2ag_split_bf16x3': [vp, ci, ci, ci, vp, vp],
    's2ag_gemm_split_fwd': [vp, vp, vp, vp, ci, ci, ci, ci, vp],
    's2ag_conv1d_nlc_fwd_split': [vp, vp, vp, vp, PG, PE, vp, vp, vp],
    's2ag_split_bf16x3_t': [vp, ci, ci, ci, ci, ci, vp, vp, vp],
    's2ag_gemm_split_acc': [vp, vp, vp, ci, ci, ci, ci, vp],
    's2ag_audio_decode': [vp, vp, vp, ci, ci, vp],
    's2ag_to_f32': [vp, ci, vp, cll, vp],
    's2ag_gru_coop_supported': [ci],
    's2ag_gru_coop_split_pieces': [],
    's2ag_gru_coop_fwd_slices': [ci],
    's2ag_gru_coop_set_split_pieces': [ci],
    's2ag_gru_coop_split_override': [],
    's2ag_gru_coop_workspace_bytes': [ci, ci, ci, ci],
    's2ag_gru_coop_fwd': [vp, vp, vp, vp, vp, vp, ci, ci, ci, PE, vp, vp],
    's2ag_gru_coop_fwd_multi_supported': [ci, ci, ci],
    's2ag_gru_coop_fwd_multi_workspace_bytes': [ci, ci, ci, ci],
    's2ag_gru_coop_fwd_multi': [ci, vp, vp, vp, vp, vp, vp, ci, ci, ci, cf, vp, cu, vp, vp],
    's2ag_gru_coop_fwd_multi_error_word_offset': [ci, ci, ci, ci, C.POINTER(cll)],
    's2ag_gru_coop_bwd': [vp, ci, ci, vp, vp, vp, vp, vp, ci, ci, ci, PE, vp, vp],
    's2ag_gru_coop_error_word_offset': [ci, ci, ci, ci, C.POINTER(cll)],
    's2ag_gru_coop_set_error_flag': [vp],
    's2ag_calib_traffic': [vp, cll, ci, vp],
    's2ag_bf16_conv_stats_rows': [ci],
    's2ag_bf16_conv': [vp, PE, vp, vp, vp],
    's2ag_bf16_conv_wgrad': [vp, vp],
    's2ag_bf16_conv_wgrad_scratch_floats': [vp],
    's2ag_bf16_conv_wgrad_split': [vp, vp, cll, vp],
    's2ag_bf16_pack_weights': [vp, ci, vp],
    's2ag_bf16_cast': [vp, ci, cll, ci, vp, ci, ci, vp],
    's2ag_bf16_bn_apply': [vp, cll, ci, ci, vp, vp, cf, vp, vp],
    's2ag_bf16_bn_bwd': [vp, vp, cll, ci, ci, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp, vp, vp],
    's2ag_bf16_add_act': [vp, vp, cll, cf, vp, vp],
    's2ag_bf16_epilogue_bwd': [vp, vp, cll, ci, ci, PE, vp, vp],
    's2ag_bf16_embedding_fwd': [vp, vp, cll, ci, ci, vp, ci, PE, vp],
    's2ag_bf16_embedding_bwd': [vp, vp, ci, cll, ci, ci, vp, PE, vp],
    's2ag_bf16_conv_c1_fwd': [vp, vp, vp, vp, PG, vp, vp, vp],
    's2ag_bf16_conv_c1_wgrad': [vp, vp, vp, vp, PG, vp],
    's2ag_bf16_conv_c1_rows': [PG],
    's2ag_wave_fwd_rows': [ci, ci, ci, ci],
    's2ag_wave_conv_fwd': [vp, vp, vp, cf, vp, ci, vp, vp, ci, vp, vp, ci, ci, ci, ci, ci, vp],
    's2ag_wave_conv1_fwd_rows': [PG],
    's2ag_wave_conv1_fwd': [vp, vp, vp, vp, PG, vp, vp, vp],
    's2ag_wave_dgrad_rows': [ci, ci, ci],
    's2ag_wave_conv_dgrad': [vp, vp, vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp],
    's2ag_wave_wgrad_blocks': [ci, ci, ci, ci],
    's2ag_wave_conv_wgrad': [vp, vp, vp, vp, vp, ci, vp, vp, vp, cf, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp],
    's2ag_wave_bn_bwd_fold': [vp, ci, ci, cll, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    's2ag_wave12_pack_elems': [],
    's2ag_wave12_pack': [vp, vp, vp, vp],
    's2ag_wave12_stats_rows': [ci, ci],
    's2ag_wave12_stats': [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp],
    's2ag_wave12_act_signs': [vp, vp, vp, vp, vp, ci, vp, ci, ci, ci, ci, vp],
    's2ag_wave12_fwd_rows': [ci, ci],
    's2ag_wave12_fwd': [vp, vp, vp, vp, vp, cf, vp, vp, ci, vp, vp, ci, ci, ci, ci, ci, vp],
    's2ag_wave12_bwd_blocks': [ci, ci, ci],
    's2ag_wave12_set_bwd_block_cap': [ci],
    's2ag_wave12_set_trace': [vp],
    's2ag_wave12_bwd': [vp, vp],
    's2ag_wave_conv1_wgrad_blocks': [PG],
    's2ag_wave_conv1_wgrad': [vp, vp, vp, vp, vp, vp, vp, vp, vp, PG, vp],
    's2ag_bf16_tcn_clips_per_block': [ci, ci, ci],
    's2ag_bf16_tcn_pack_elems': [ci],
    's2ag_bf16_tcn_sign_bytes': [ci, ci],
    's2ag_bf16_tcn_keep_bytes': [ci, ci, ci],
    's2ag_bf16_tcn_pack': [vp, ci, ci, vp, vp],
    's2ag_bf16_tcn_fwd': [vp, vp],
    's2ag_bf16_tcn_set_trace': [vp],
    's2ag_bf16_tcn_bwd': [vp, vp],
    's2ag_bf16_conv_wgrad_multi': [vp, ci, vp],
    's2ag_tcn32_supported': [ci, ci, ci],
    's2ag_tcn32_pack_elems': [ci],
    's2ag_tcn32_keep_bytes': [ci, ci],
    's2ag_tcn32_pack': [vp, ci, ci, vp, vp],
    's2ag_tcn32_fwd': [vp, vp],
    's2ag_tcn32_fwd_passes': [vp, ci, vp, ci, vp],
    's2ag_tcn32_bwd': [vp, vp],
    's2ag_bf16_conv_wgrad_tr_scratch_floats': [vp, ci],
    's2ag_bf16_conv_wgrad_tr': [vp, ci, vp, cll, vp],
    's2ag_f32_wgrad_tr_scratch_floats': [vp, ci],
    's2ag_f32_wgrad_tr': [vp, ci, vp, cll, vp],
    's2ag_f32_wgrad_tr_scratch_floats_n': [vp, ci, ci],
    's2ag_f32_wgrad_tr_n': [vp, ci, vp, cll, ci, vp],
    's2ag_wgrad_tr_set_trace': [vp],
    's2ag_rows_unique': [vp, ci, ci, ci, vp, vp, vp, vp, vp],
    's2ag_rows_pack': [vp, vp, ci, ci, ci, vp, vp],
    's2ag_rows_merge': [vp, ci, ci, ci, ci, vp, vp],
    's2ag_reparam_fwd': [vp, vp, ci, vp, cu, vp, vp],
    's2ag_reparam_bwd': [vp, vp, ci, vp, cu, vp, vp, vp],
    's2ag_dis_loss': [vp, vp, ci, vp, vp, vp, vp],
    's2ag_gen_loss': [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, C.POINTER(cf), vp, vp, vp, vp, vp, vp, vp],
    's2ag_adam_step': [vp, vp, vp, vp, cll, cf, cf, cf, cf, vp, cf, vp],
    's2ag_pose_metrics': [vp, vp, vp, ci, ci, ci, vp, vp],
    's2ag_adam_set_guard': [vp],
    's2ag_counter_inc': [vp, vp, vp],
    's2ag_rng_snapshot': [vp, vp, vp],
    's2ag_rng_snapshots': [vp, vp, ci, vp, ci, vp],
    's2ag_make_pre_seq': [vp, vp, ci, ci, ci, ci, vp],
    's2ag_concat_cols': [vp, vp, vp, vp, ci, vp, cll, ci, vp],
    's2ag_sum_frames': [vp, ci, ci, ci, ci, ci, vp, vp],
    's2ag_dropout_mask': [vp, cu, cf, cll, vp, vp],
    's2ag_normal_noise': [vp, cu, cll, vp, vp],
}

_lib = None


class S2AGLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and type-check the shared library.  Raises S2AGLibraryError loudly if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    from . import config
    path = config.get('HIP_LIB') or LIB_PATH     # override: another build of the same ABI (debug / asan flavour)
    if not os.path.exists(path):
        raise S2AGLibraryError(
            f'{path} not found: the S2AG HIP kernels are not built. Run '
            f'`python -m speech2affective_gestures_amd.build` (hipcc, gfx950). There is no CPU fallback.')
    lib = C.CDLL(path)
    for name, args in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise S2AGLibraryError(f'{path} lacks symbol {name}; rebuild it') from e
        fn.argtypes = args
        fn.restype = cll if name in ('s2ag_gru_coop_workspace_bytes', 's2ag_gru_coop_fwd_multi_workspace_bytes', 's2ag_bf16_tcn_pack_elems', 's2ag_bf16_tcn_sign_bytes', 's2ag_bf16_tcn_keep_bytes',
                              's2ag_bf16_conv_wgrad_scratch_floats', 's2ag_bf16_conv_wgrad_tr_scratch_floats',
                              's2ag_f32_wgrad_tr_scratch_floats', 's2ag_f32_wgrad_tr_scratch_floats_n', 's2ag_tcn32_pack_elems', 's2ag_tcn32_keep_bytes') else ci
    if lib.s2ag_abi_version() != ABI_VERSION:
        raise S2AGLibraryError(f'ABI version mismatch: _lib.py binds version {ABI_VERSION} of include/s2ag_hip.h, {path} is '
                               f'version {lib.s2ag_abi_version()}; rebuild it')
    if config.get('CRASH_TRACE'):     # native back trace on SIGSEGV & co (csrc/debug.hip)
        lib.s2ag_install_crash_handler(2)
    config.push_to_library(lib)        # the library never reads the environment: options come from the registry
    _lib = lib
    return lib


ABI_VERSION = 2             # S2AG_ABI_VERSION (include/s2ag_hip.h); bump both with every signature / struct change
E_UNSUPPORTED = -2          # S2AG_E_UNSUPPORTED (include/s2ag_hip.h)


def require_gpu_device(device, what: str):
    """The one statement of 'no CPU fallback' for state that lives on a device (noise counters, parameter arenas)."""
    import torch
    device = torch.device(device)
    if device.type != 'cuda':
        raise RuntimeError(f'{what} lives on the GPU; there is no CPU fallback')
    return device


def check(rc: int, what: str):
    if rc != 0:
        kind = {-1: 'bad argument', -2: 'unsupported shape'}.get(rc, f'hipError {rc}')
        raise RuntimeError(f'{what} failed: {kind}')
