"""ctypes binding of libdensebox_hip.so (the C ABI declared in include/densebox_hip.h).

The library must be present (built in-tree by ``densebox_amd._build`` /
``__graft_entry__.build()``); there is no fallback -- a missing library raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DBX_LIB: load another build of the same library (tools/build_variant.sh makes same-box A/B builds of one source file)
LIB_PATH = os.environ.get('DBX_LIB') or os.path.join(_HERE, 'csrc', 'libdensebox_hip.so')

F16, BF16, F32 = 0, 1, 2
DTYPE_ID = {'f16': F16, 'bf16': BF16, 'f32': F32}
ESIZE = {F16: 2, BF16: 2, F32: 4}

EPI_BIAS, EPI_RELU, EPI_GATE, EPI_DROPMASK, EPI_ACCUM, EPI_F32_NCHW, EPI_DROPHASH = 1, 2, 4, 8, 16, 32, 64
CONV_WFRAG = 128          # w_packed is in MFMA-fragment order (pack modes 4/5)
K_IGEMM, K_DMA, K_BAND, K_C64, K_C8, K_WS, K_P8 = 1, 2, 3, 4, 5, 6, 7


class View(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('n', C.c_int32), ('h', C.c_int32), ('w', C.c_int32),
                ('pad', C.c_int32), ('ld', C.c_int32), ('c_off', C.c_int32), ('c', C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [('dtype', C.c_int32), ('kh', C.c_int32), ('kw', C.c_int32), ('cpad', C.c_int32),
                ('cin_pad', C.c_int32), ('cout_pad', C.c_int32), ('epilogue', C.c_int32), ('drop_seed', C.c_uint32)]


class ConvPlan(C.Structure):
    _fields_ = [('kernel', C.c_int32), ('tile_m', C.c_int32), ('tile_n', C.c_int32), ('w_frag', C.c_int32),
                ('name', C.c_char * 64)]


class LossDesc(C.Structure):
    _fields_ = [('kind', C.c_int32), ('n', C.c_int32), ('half_neg', C.c_int32),
                ('lambda_loc', C.c_float), ('lambda_det', C.c_float), ('lambda_lm', C.c_float),
                ('use_labels', C.c_int32)]


class LossIO(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in (
        'bbox', 'vertices', 'labels', 'rand_neg', 'lm_rand_neg',
        'score', 'loc', 'lm', 'rf', 'lmloc',
        'd_score', 'd_loc', 'd_lm', 'd_rf', 'd_lmloc',
        'loss', 'mask_cls', 'mask_lm', 'neg_idx', 'lm_neg_idx', 'pos_count')]


_VP, _I32, _I64, _F, _D = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
_PV, _PC = C.POINTER(View), C.POINTER(ConvDesc)

# name -> (restype, argtypes).  Must list every symbol include/densebox_hip.h declares
# (tests/test_abi.py cross-checks this table against the header and the built library).
SIGNATURES = {
    'dbx_last_error': (C.c_char_p, []),
    'dbx_version': (C.c_int, []),
    'dbx_device_arch': (C.c_int, [C.c_int]),
    'dbx_conv_packed_elems': (_I64, [_PC]),
    'dbx_conv_forward': (C.c_int, [_PC, _PV, _VP, _VP, _PV, _PV, _VP, _I32, _VP]),
    'dbx_conv_plan': (C.c_int, [_PC, _PV, _PV, C.POINTER(ConvPlan)]),
    'dbx_conv_forward_split': (C.c_int, [_PC, _PV, _VP, _VP, _PV, _PV, _PV, _PV, _I32, _I32, _VP]),
    'dbx_heads_forward_fusable': (C.c_int, [_PC, _PV, _PV, C.POINTER(C.c_int32), _I32]),
    'dbx_heads_forward_fused_scratch_bytes': (_I64, [_I32, _I64]),
    'dbx_heads_forward_fused': (C.c_int, [_PC, _PV, _VP, _VP, _PV, _VP, _VP, C.POINTER(C.c_int32), _I32, _VP, _VP, _VP]),
    'dbx_heads_forward_fused_heads': (C.c_int, [_PC, _PV, _VP, _VP, _PV, _VP, _VP, C.POINTER(C.c_int32), _I32, C.POINTER(C.c_void_p), _VP, _VP]),
    'dbx_dp_unique_id': (C.c_int, [_VP]),
    'dbx_dp_init': (C.c_int, [_VP, _I32, _I32, C.POINTER(C.c_void_p)]),
    'dbx_dp_allreduce_sum_f32': (C.c_int, [_VP, _VP, _I64, _VP]),
    'dbx_dp_destroy': (C.c_int, [_VP]),
    'dbx_conv_dgrad_wgrad1_scratch_bytes': (_I64, []),
    'dbx_conv_dgrad_wgrad1_fusable': (C.c_int, [_PC, _PV, _PV, _PV]),
    'dbx_conv_dgrad_wgrad1': (C.c_int, [_PC, _PV, _VP, _PV, _PV, _I32, _VP, _VP, _VP, _I32, _VP]),
    'dbx_conv_pool_fusable': (C.c_int, [_PC, _PV, _PV]),
    'dbx_conv_forward_pool': (C.c_int, [_PC, _PV, _VP, _VP, _PV, _PV, _I32, _VP]),
    'dbx_conv_forward_pool_idx': (C.c_int, [_PC, _PV, _VP, _VP, _PV, _PV, _I32, _VP, _VP]),
    'dbx_pack_weight': (C.c_int, [_I32, _I32, _VP, _I32, _I32, _I32, _I32, _VP, _I32, _I32, _I32, _I32, _VP]),
    'dbx_fold_heads': (C.c_int, [_VP, _VP, _VP, _VP, _I32, _VP, _VP, _VP]),
    'dbx_fold_refine': (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _VP, _VP, _VP, _VP]),
    'dbx_refine_eval': (C.c_int, [_VP, _VP, _I32, _I32, _I32, _VP, _VP, _VP, _VP]),
    'dbx_refine_backward_scratch_bytes': (C.c_int64, [_I32, _I32, _I32]),
    'dbx_refine_backward': (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    'dbx_upsample_bilinear_nchw_f32': (C.c_int, [_VP, _I32, _I32, _I32, _VP, _I32, _I32, _VP]),
    'dbx_pack_multi': (C.c_int, [_I32, _VP, _I32, _I64, _VP]),
    'dbx_sgd_pack_step': (C.c_int, [_I32, _VP, _I32, _I64, _VP, _F, _F, _F, _I32, _VP]),
    'dbx_head2_dgrad': (C.c_int, [_I32, _PV, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), _I32, _PV, _VP, _I32, _I32, C.c_uint32, _VP]),
    'dbx_head2_wgrad_scratch_bytes': (_I64, [_I32, _I32]),
    'dbx_head2_wgrad': (C.c_int, [_I32, _PV, _PV, C.POINTER(C.c_int32), _I32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _VP, _VP]),
    'dbx_head2_backward': (C.c_int, [_I32, _PV, _PV, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), _I32, _PV, _VP, _I32, _I32, C.c_uint32,
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _VP, _VP]),
    'dbx_head2_backward_up': (C.c_int, [_I32, _PV, _PV, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), _I32, _PV, _VP, _I32, _I32, C.c_uint32,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _VP, _PV, _VP]),
    'dbx_perspective_matrix': (C.c_int, [_VP, _VP, _VP]),
    'dbx_warp_perspective_u8': (C.c_int, [_VP, _I32, _I32, _I32, _VP, _VP, _I32, _I32, _VP]),
    'dbx_conv_wgrad_scratch_bytes': (_I64, [_I32, _PV, _PV, _I32, _I32]),
    'dbx_conv_wgrad': (C.c_int, [_I32, _PV, _PV, _I32, _I32, _I32, _I32, _I32, _VP, _VP, _VP, _I32, _VP]),
    'dbx_conv_wgrad_slice': (C.c_int, [_I32, _PV, _PV, _I32, _I32, _I32, _I32, _I32, _VP, _I32, _I32, _VP, _VP, _I32, _VP]),
    'dbx_conv_wgrad_pool_dz_ok': (C.c_int, [_I32, _PV, _PV, _I32, _I32]),
    'dbx_conv_wgrad_pool_dz': (C.c_int, [_I32, _PV, _VP, _I32, _PV, _PV, _I32, _I32, _I32, _I32, _I32, _VP, _VP, _VP, _I32, _I32, _VP]),
    'dbx_head2_backward_up_fused': (C.c_int, [_I32, _PV, _PV]),
    'dbx_heads1_wgrad_gen_ok': (C.c_int, [_I32, _PV, _I32]),
    'dbx_heads1_dgrad_gen': (C.c_int, [_I32, _PV, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), _I32, _I32, C.c_uint32, _VP, _PV, _PV, _VP]),
    'dbx_heads1_wgrad_gen': (C.c_int, [_I32, _PV, _PV, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), _I32, _I32, C.c_uint32, _I32, _VP, _I32, _I32,
                             _VP, _VP, _VP]),
    'dbx_conv_wgrad_plan': (C.c_int, [_I32, _PV, _PV, _I32, _I32, C.c_char_p, _I32, C.POINTER(C.c_int32)]),
    'dbx_nchw_to_framed': (C.c_int, [_I32, _VP, _I32, _PV, _VP]),
    'dbx_u8hwc_to_framed': (C.c_int, [_I32, _VP, _PV, C.POINTER(C.c_float), C.POINTER(C.c_float), _VP]),
    'dbx_nchw_to_framed_ch': (C.c_int, [_I32, _VP, _I32, _PV, _I32, _VP]),
    'dbx_nchw_to_framed_slots': (C.c_int, [_I32, _VP, _VP, _I32, _I32, _PV, _VP]),
    'dbx_framed_to_nchw_f32': (C.c_int, [_I32, _PV, _VP, _VP]),
    'dbx_framed_add_ch': (C.c_int, [_I32, _PV, _I32, _I32, _PV, _I32, _VP]),
    'dbx_maxpool2x2': (C.c_int, [_I32, _PV, _PV, _VP]),
    'dbx_maxpool2x2_bwd': (C.c_int, [_I32, _PV, _PV, _PV, _I32, _I32, _VP]),
    'dbx_maxpool_idx_bytes': (C.c_int64, [_I32, _I32, _I32, _I32]),
    'dbx_maxpool2x2_idx': (C.c_int, [_I32, _PV, _PV, _VP, _VP]),
    'dbx_maxpool2x2_bwd_idx': (C.c_int, [_I32, _VP, _PV, _PV, _I32, _I32, _VP]),
    'dbx_upsample_bilinear': (C.c_int, [_I32, _PV, _PV, _VP]),
    'dbx_upsample_bilinear_bwd': (C.c_int, [_I32, _PV, _PV, _PV, _VP]),
    'dbx_loss_scratch_bytes': (C.c_int64, [_I32]),
    'dbx_loss_forward_backward': (C.c_int, [C.POINTER(LossDesc), C.POINTER(LossIO), _VP, _VP]),
    'dbx_count_positives': (C.c_int, [_VP, _VP, _I32, _VP, _VP]),
    'dbx_init_score_map': (C.c_int, [_VP, _VP, _I32, _VP, _VP]),
    'dbx_init_offset_map': (C.c_int, [_VP, _VP, _I32, _I32, _VP, _VP]),
    'dbx_init_lm_heatmap': (C.c_int, [_VP, _VP, _I32, _I32, _VP, _VP]),
    'dbx_mask_by_sel': (C.c_int, [_VP, _I32, _VP, _I64, _VP, _I32, _VP]),
    'dbx_mask_gray_zone_cls': (C.c_int, [_VP, _VP, _VP, _I32, _VP]),
    'dbx_mask_gray_zone_lm': (C.c_int, [_VP, _I32, _VP, _I64, _VP]),
    'dbx_sgd_step': (C.c_int, [_VP, _VP, _I32, _I64, _F, _F, _F, _I32, _VP]),
    'dbx_grad_guard': (C.c_int, [_VP, _I64, _VP, _I32, _VP]),
    'dbx_sgd_step_guarded': (C.c_int, [_VP, _VP, _I32, _I64, _F, _F, _F, _I32, _VP, _I32, _VP]),
    'dbx_sgd_pack_step_guarded': (C.c_int, [_I32, _VP, _I32, _I64, _VP, _F, _F, _F, _I32, _VP, _I32, _VP]),
    'dbx_detect_scratch_bytes': (_I64, [_I32, _I32, _I32]),
    'dbx_detect': (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _D, _VP, _I32, _VP, _VP, _VP, _VP]),
    'dbx_nms': (C.c_int, [_VP, _I32, _I32, _D, _VP, _VP, _VP]),
}

ABI_VERSION = 7          # include/densebox_hip.h DBX_ABI_VERSION this binding was written against
_lib = None
MISSING = []


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libdensebox_hip.so is not built (%s). Run `python -c "import __graft_entry__ as g; '
                               'g.build()"`; densebox_amd has no fallback path.' % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError:         # header/library out of sync: tests/test_abi.py fails on any entry here
                MISSING.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
        if 'dbx_version' not in MISSING and L.dbx_version() != ABI_VERSION:
            raise RuntimeError('libdensebox_hip.so at %s has ABI version %d, this binding needs %d (struct layouts / scratch '
                               'contracts differ): rebuild it' % (LIB_PATH, L.dbx_version(), ABI_VERSION))
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError('libdensebox_hip: %s (status %d)' % (lib().dbx_last_error().decode(), rc))


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
