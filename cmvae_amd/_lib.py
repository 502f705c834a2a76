"""ctypes loader for libcmvae_hip.so (the C-ABI declared in include/cmvae_abi.h).

The product path has no fallback: if the library is missing or a symbol is absent this
module raises at import time.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CMVAE_HIP_LIB') or os.path.join(_HERE, 'libcmvae_hip.so')      # override: profiling builds

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_float = ctypes.c_float
c_size_t = ctypes.c_size_t


class CausalDesc(ctypes.Structure):
    """struct cmvae_causal (include/cmvae_abi.h)."""
    _fields_ = [('linear', ctypes.c_int32), ('W', c_void_p), ('b1', c_void_p), ('w2', c_void_p), ('b2', c_void_p)]


_PC = ctypes.POINTER(CausalDesc)


class ConvGeom(ctypes.Structure):
    """struct cmvae_conv_geom (include/cmvae_abi.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in ('rows_total', 'Hg', 'Wg', 'Hi', 'Wi', 'sy', 'sx', 'Ho', 'Wo', 'oys',
                                              'oxs', 'oy0', 'ox0', 'ntaps', 'a_pitch', 'o_pitch', 'w_pitch')] + \
               [('w_nblk_stride', ctypes.c_int64),
                ('dy', ctypes.c_int32 * 32), ('dx', ctypes.c_int32 * 32), ('a_col', ctypes.c_int32 * 32),
                ('w_off', ctypes.c_int32 * 32), ('pixel_major', ctypes.c_int32)]


_PG = ctypes.POINTER(ConvGeom)
c_int64 = ctypes.c_int64

# name -> (restype, argtypes); must list every symbol of include/cmvae_abi.h
SIGNATURES = {
    'cmvae_last_error': (ctypes.c_char_p, []),
    'cmvae_abi_version': (c_int, []),
    'cmvae_causal_rows': (c_int, [c_void_p, c_int, _PC, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'cmvae_gram_inverse': (c_int, [c_void_p, c_float, c_int, _PC, c_void_p, c_void_p]),
    'cmvae_unsup_prior_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'cmvae_unsup_prior': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, _PC, c_int, c_int, c_float,
                                  c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'cmvae_unsup_prior_noalloc_max_bytes': (c_size_t, []),
    'cmvae_unsup_prior_noalloc': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, _PC, c_int, c_int, c_float,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'cmvae_em_pass': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t,
                              ctypes.POINTER(c_int), c_void_p]),
    'cmvae_semi_prior_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'cmvae_semi_prior': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, _PC, c_int, c_float,
                                 c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'cmvae_posterior_argmax': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p]),
    'cmvae_mixture_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'cmvae_mixture_loglik_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cmvae_mixture_loglik_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                         c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    'cmvae_center_rows_chunks': (c_int, [c_int]),
    'cmvae_center_rows': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'cmvae_reparam_logq_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'cmvae_reparam_logq_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p]),
    'cmvae_expm_trace': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'cmvae_elbo_tail_blocks': (c_int, []),
    'cmvae_elbo_tail_fwd': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                    c_int64, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    'cmvae_elbo_tail_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int64, c_int64,
                                    c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    'cmvae_prior_logw_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'cmvae_prior_logw_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'cmvae_colsum_reduce': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'cmvae_bn_finalize': (c_int, [c_void_p, ctypes.c_double, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'cmvae_bn_finalize_part': (c_int, [c_void_p, c_int, ctypes.c_double, c_void_p, c_void_p, c_float, c_float, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'cmvae_bn_bwd_finalize_part': (c_int, [c_void_p, c_int, ctypes.c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p]),
    'cmvae_bn_eval_fold': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                   c_void_p]),
    'cmvae_rowpass_grid': (c_int, [c_int64]),
    'cmvae_bn_relu_bwd_reduce': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                         c_void_p, c_void_p]),
    'cmvae_bn_bwd_finalize': (c_int, [c_void_p, c_void_p, ctypes.c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    'cmvae_bn_relu_bwd_apply': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    'cmvae_dec_out_grid': (c_int, [c_int]),
    'cmvae_dec_out_bwd_partial_len': (c_int, []),
    'cmvae_dec_out_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                  c_void_p, c_void_p, c_void_p]),
    'cmvae_dec_out_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int, c_int, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    'cmvae_dec_out_bwd_sums_partial_len': (c_int, []),
    'cmvae_dec_out_bwd_sums_grid': (c_int, [c_int]),
    'cmvae_dec_out_bwd_sums': (c_int, [c_void_p] * 8 + [c_int, c_int, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                                       c_void_p]),
    'cmvae_dec_out_bwd_apply': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int, c_int, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    'cmvae_enc_conv1_pool_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                         c_void_p]),
    'cmvae_enc_conv1_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'cmvae_enc_conv1_bwd_lin_cols': (c_int, []),
    'cmvae_enc_conv1_tap_sums': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cmvae_enc_conv1_bwd_lin_rc_grid': (c_int, [c_int]),
    'cmvae_enc_conv1_bwd_lin_rc': (c_int, [c_void_p] * 9 + [c_int, c_int, c_int, c_void_p, c_void_p]),
    'cmvae_enc_conv1_lin_finalize': (c_int, [c_void_p, c_void_p, c_float, c_int, c_int, c_int] + [c_void_p] * 11),
    'cmvae_pool_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cmvae_pool_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                               c_void_p, c_void_p, c_void_p]),
    'cmvae_conv_gemm_grid': (c_int, [c_int, c_int]),
    'cmvae_conv_gemm': (c_int, [_PG, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                c_void_p, c_void_p]),
    'cmvae_conv_gemm_bnbwd': (c_int, [_PG] + [c_void_p] * 10),
    'cmvae_conv_gemm_multi_grid': (c_int, []),
    'cmvae_conv_bf16_pipe_min_rows': (c_int, []),
    'cmvae_conv_gemm_multi': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p]),
    'cmvae_conv_wgrad_workspace_bytes': (c_size_t, [c_int]),
    'cmvae_conv_wgrad': (c_int, [_PG, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                 ctypes.POINTER(c_int64), c_int64, c_int64, c_void_p, c_void_p, c_int,
                                 c_void_p, c_size_t, c_void_p]),
    'cmvae_conv_wgrad_multi': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                       ctypes.POINTER(c_int64), c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                       c_size_t, c_void_p]),
    'cmvae_linear_fwd': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                 c_void_p]),
    'cmvae_linear_mse_partials': (c_int, [c_int, c_int]),
    'cmvae_linear_mse_fwd': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                     c_void_p, c_int, c_void_p, c_void_p]),
    'cmvae_mse_act_bwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_float, c_void_p,
                                  c_void_p]),
    'cmvae_act_bwd': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    'cmvae_linear_bwd_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'cmvae_linear_bwd': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                 c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'cmvae_causal_lc_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'cmvae_causal_lc_bwd_grid': (c_int, [c_int]),
    'cmvae_causal_lc_bwd_partial_len': (c_int, []),
    'cmvae_causal_lc_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'cmvae_attention_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                    c_void_p]),
    'cmvae_attention_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'cmvae_attention_qkv_fwd': (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    'cmvae_attention_qkv_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                                        c_void_p, c_void_p]),
    'cmvae_adam_step_dev': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_float, c_float,
                                    c_float, c_float, c_float, c_void_p]),
    'cmvae_adam_step': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_float, c_float, c_float,
                                c_float, c_float, c_void_p]),
    'cmvae_em_seed_draw_host': (c_int, [c_void_p, ctypes.POINTER(c_int), c_int, c_int, c_int, c_void_p, c_void_p]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'cmvae_amd: %s not found. Build it with `make -C cmvae_amd/csrc` (or '
            '`python -c "import __graft_entry__ as g; g.build()"`). There is no fallback path.' % LIB_PATH)
    # PyTorch-ROCm ships its own libamdhip64: import it first so that the library below binds to the SAME HIP
    # runtime as the tensors it is handed (loaded the other way round, the process ends up with two runtimes and
    # every launch from here fails with "no ROCm-capable device is detected").
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc, what=''):
    if rc != 0:
        msg = lib.cmvae_last_error()
        raise RuntimeError('%s failed: %s' % (what or 'cmvae call', msg.decode() if msg else 'unknown error'))
