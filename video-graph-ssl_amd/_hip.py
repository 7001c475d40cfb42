"""ctypes binding of libgca_hip.so (the C ABI declared in include/gca_hip.h).

The product path has NO CPU fallback: importing this module without the built library, or
calling any op without a GPU tensor, raises.  Build with ``python -c "import __graft_entry__ as
g; g.build()"`` (or ./build_hip.sh) -- hipcc cross-compiles gfx950 without a GPU.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GCA_LIB_PATH') or os.path.join(_HERE, 'libgca_hip.so')      # (GCA_LIB_PATH: A/B runs against another build)


class HipLibraryMissing(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise HipLibraryMissing(
        'libgca_hip.so is not built (%s). Run ./build_hip.sh or __graft_entry__.build(); there is no '
        'CPU fallback for the product path.' % LIB_PATH)
lib = C.CDLL(LIB_PATH)

c_i32, c_i64, c_f32, c_f64, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p


class ConvGeom(C.Structure):
    _fields_ = [(n, c_i32) for n in ('N', 'C', 'D', 'H', 'W', 'K', 'kd', 'kh', 'kw', 'sd', 'sh', 'sw',
                                     'pd', 'ph', 'pw', 'OD', 'OH', 'OW')] + [('x_batch_stride', c_i64)] + \
               [(n, c_i32) for n in ('tune_fwd_bm', 'tune_fwd_splits', 'tune_dgrad_bm', 'tune_dgrad_splits',
                                     'tune_wgrad_splits', 'tune_wgrad_tile', 'tune_fwd_tail', 'tune_dgrad_tail',
                                     'tune_fwd_math', 'tune_dgrad_math', 'tune_wgrad_math', 'tune_fwd_box',
                                     'tune_dgrad_box', 'act_f16')]


class PoolGeom(C.Structure):
    _fields_ = [(n, c_i32) for n in ('N', 'C', 'D', 'H', 'W', 'kd', 'kh', 'kw', 'sd', 'sh', 'sw',
                                     'pd', 'ph', 'pw', 'OD', 'OH', 'OW')]


_GP, _PP = C.POINTER(ConvGeom), C.POINTER(PoolGeom)
# name -> (restype, argtypes): exactly the declarations of include/gca_hip.h
SIGNATURES = {
    'gca_version': (c_i32, []),
    'gca_conv_halo_occupancy': (c_i32, [c_i32, c_i32, c_i32, c_i64]),
    'gca_set_conv_math': (c_i32, [c_i32]),
    'gca_get_conv_math': (c_i32, []),
    'gca_conv_pack_elems': (c_i64, [_GP, c_i32]),
    'gca_conv_pack': (c_i32, [_GP, c_i32, c_vp, c_vp, c_vp]),
    'gca_conv_table_rows': (c_i64, [_GP, c_i32]),
    'gca_conv_table_build_host': (c_i32, [_GP, c_i32, c_vp]),
    'gca_conv_fwd_stat_parts': (c_i64, [_GP]),
    'gca_conv_kernel_cfg': (c_i32, [_GP, c_i32, c_vp]),
    'gca_conv_pack_layout': (c_i64, [_GP, c_i32]),
    'gca_conv_fwd_ws_bytes': (c_i64, [_GP]),
    'gca_conv_fwd': (c_i32, [_GP, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'gca_conv_dgrad_ws_bytes': (c_i64, [_GP]),
    'gca_conv_dgrad': (c_i32, [_GP, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    'gca_conv_wgrad_cfg': (c_i32, [_GP, c_vp]),
    'gca_conv_wgrad_ws_bytes': (c_i64, [_GP]),
    'gca_conv_wgrad': (c_i32, [_GP, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    'gca_bias_grad': (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i32, c_vp]),
    'gca_bn_stats': (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_i32, c_vp]),
    'gca_bn_stats_parts': (c_i64, [c_i64, c_i64, c_i64]),
    'gca_bn_finalize': (c_i32, [c_vp, c_vp, c_i64, c_i64, c_f64, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp,
                                c_vp, c_vp, c_vp, c_vp, c_vp]),
    'gca_bn_train_fwd': (c_i32, [c_vp, c_vp, c_i64, c_i64, c_f64, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp,
                                 c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i64, c_vp, c_i64, c_i32, c_vp]),
    'gca_bn_fold_eval': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_f32, c_i64, c_vp, c_vp, c_vp]),
    'gca_bn_apply': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i64, c_i64, c_vp, c_i64, c_i32, c_vp]),
    'gca_bn_bwd_ws_bytes': (c_i64, [c_i64, c_i64, c_i64]),
    'gca_bn_bwd': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp,
                           c_vp, c_i32, c_i64, c_vp, c_vp, c_vp, c_i32, c_vp]),
    'gca_maxpool3d_fwd': (c_i32, [_PP, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    'gca_maxpool3d_bwd': (c_i32, [_PP, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    'gca_avgpool3d_fwd': (c_i32, [_PP, c_vp, c_vp, c_vp]),
    'gca_avgpool3d_bwd': (c_i32, [_PP, c_vp, c_vp, c_i32, c_vp]),
    'gca_wavgpool_fwd': (c_i32, [c_vp, c_vp, c_f32, c_i64, c_i64, c_i64, c_vp, c_i32, c_vp]),
    'gca_wavgpool_bwd': (c_i32, [c_vp, c_vp, c_f32, c_i64, c_i64, c_i64, c_vp, c_i32, c_vp]),
    'gca_relu_fwd': (c_i32, [c_vp, c_i64, c_vp, c_vp]),
    'gca_relu_bwd': (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    'gca_l2norm_fwd': (c_i32, [c_vp, c_i64, c_i64, c_f32, c_vp, c_vp, c_vp]),
    'gca_l2norm_bwd': (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]),
    'gca_negcos_fwd_bwd': (c_i32, [c_vp, c_vp, c_i64, c_i64, c_f32, c_vp, c_i32, c_vp, c_vp]),
    'gca_infonce_ws_bytes': (c_i64, [c_i64, c_i64]),
    'gca_moco_logits_fwd': (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'gca_nce_softmax_loss_fwd': (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp]),
    'gca_nce_softmax_loss_bwd': (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp, c_f32, c_vp, c_vp]),
    'gca_moco_logits_bwd': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_i64, c_i64, c_i64, c_f32,
                                    c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    'gca_queue_enqueue': (c_i32, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp]),
    'gca_queue_advance': (c_i32, [c_vp, c_i64, c_i64, c_vp]),
    'gca_graph_gram_ws_bytes': (c_i64, [c_i64, c_i64, c_i64, c_i64]),
    'gca_graph_adj_fwd': (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i32, c_f32, c_f32, c_vp, c_vp,
                                  c_vp, c_vp, c_vp, c_vp]),
    'gca_graph_adj_bwd': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i32, c_f32,
                                  c_f32, c_vp, c_vp, c_vp]),
    'gca_graph_gcn_fwd': (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp]),
    'gca_graph_gcn_bwd_ws_bytes': (c_i64, [c_i64, c_i64, c_i64, c_i64]),
    'gca_graph_gcn_bwd': (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp]),
    'gca_ema_update': (c_i32, [c_vp, c_vp, c_i64, c_f32, c_vp]),
    'gca_sgd_step': (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_f32, c_f32, c_i32, c_i32, c_vp, c_vp]),
    'gca_grad_clip_ws_bytes': (c_i64, []),
    'gca_grad_clip_coef': (c_i32, [c_vp, c_i64, c_f32, c_vp, c_vp, c_vp]),
    'gca_conv_wgrad_partial': (c_i32, [_GP, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'gca_conv_xf_ok': (c_i32, [_GP]),
    'gca_conv_fwd_xf': (c_i32, [_GP, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'gca_conv_wgrad_xf': (c_i32, [_GP, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    'gca_conv_fwd_slabs': (c_i32, [_GP, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'gca_bn_train_fwd_slabs': (c_i32, [c_vp, c_i64, c_f64, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                       c_vp, c_i32, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp]),
    'gca_reduce_jobs_finalize_host': (c_i64, [c_vp, c_i64]),
    'gca_splitk_reduce_batched': (c_i32, [c_vp, c_i64, c_i64, c_vp]),
    'gca_clip_prepare': (c_i32, [c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i32, c_vp]),
    'gca_rank_ge': (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]),
    'gca_grad_unscale_clip': (c_i32, [c_vp, c_i64, c_f32, c_vp, c_f32, c_f32, c_i32, c_f32, c_vp, c_vp, c_vp]),
    'gca_scale_dev': (c_i32, [c_vp, c_i64, c_vp, c_f32, c_vp]),
    'gca_fill': (c_i32, [c_vp, c_i64, c_f32, c_vp]),
    'gca_axpy': (c_i32, [c_vp, c_vp, c_i64, c_f32, c_vp]),
    'gca_scale': (c_i32, [c_vp, c_i64, c_f32, c_vp]),
    'gca_axpy_f16': (c_i32, [c_vp, c_vp, c_i64, c_f32, c_vp]),
    'gca_cast_f16': (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_vp]),
    'gca_conv_pack_jobs_host': (c_i64, [_GP, c_i32, c_vp, c_vp, c_vp]),
    'gca_conv_pack_jobs_finalize_host': (c_i64, [c_vp, c_i64]),
    'gca_conv_pack_batched': (c_i32, [c_vp, c_i64, c_i64, c_vp]),
    'gca_gather_rows': (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp]),
}
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here = header/library drift
    _fn.restype, _fn.argtypes = _res, _args


PACK_JOB_BYTES = 128      # GCA_PACK_JOB_BYTES


class ReduceJob(C.Structure):      # gca_reduce_job
    _fields_ = [('slabs', C.c_void_p), ('dw', C.c_void_p), ('n', C.c_int64), ('splits', C.c_int32), ('accumulate', C.c_int32),
                ('first_block', C.c_int32), ('nblocks', C.c_int32)]


def ptr(t, half_ok=False):
    """Device pointer of a (contiguous, fp32/int) CUDA tensor, or NULL.  fp16 tensors are only accepted where the kernel
    behind the call takes the activation storage type as an argument (half_ok, see aptr)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('gca HIP ops need tensors on the GPU (got %s); there is no CPU fallback' % t.device)
    if t.dtype is not torch.float32 and not (half_ok and t.dtype is torch.float16) and t.is_floating_point():
        raise TypeError('this gca HIP op has no %s kernel' % t.dtype)
    return t.data_ptr()


def aptr(t):
    """Pointer of an activation tensor: fp32, or fp16 on the fp16-storage path."""
    return ptr(t, True)


def is_half(*ts):
    """1 when the activation tensors are fp16 (all of them must then be), else 0."""
    kinds = {t.dtype for t in ts if t is not None}
    if len(kinds) > 1:
        raise TypeError('mixed activation storage types: %s' % sorted(map(str, kinds)))
    return int(kinds == {torch.float16})


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed with status %d (%s)' % (what, rc, {-1: 'invalid argument', -2: 'launch error'}.get(rc, '?')))


def call(name, *args):
    check(getattr(lib, name)(*args), name)
