"""CPU: the C-ABI library loads and exports every symbol include/cmvae_abi.h declares (no compute calls:
there is no GPU here), the ctypes table covers them all, and the host surface mirrors the reference's."""
import ctypes
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'cmvae_abi.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(cmvae_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert 'cmvae_unsup_prior' in syms and 'cmvae_conv_gemm' in syms and len(syms) >= 30


def test_library_exports_every_declared_symbol():
    from cmvae_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    unbound = [s for s in declared_symbols() if s not in _lib.SIGNATURES]
    assert not unbound, unbound
    assert _lib.lib.cmvae_abi_version() >= 1


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from cmvae_amd import _lib
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(ImportError):
        _lib._load()


def test_cpu_tensors_are_rejected():
    from cmvae_amd import ops
    with pytest.raises(RuntimeError):
        ops.unsup_prior(torch.zeros(1, 8, 64), torch.zeros(1, 2, dtype=torch.int64), None, 1)
    with pytest.raises(RuntimeError):
        ops.reparam_logq(torch.zeros(1, 2, 64), torch.zeros(1, 2, 64), torch.zeros(1, 2, 3, 64))


def test_surface_matches_reference_signatures():
    """Constructor argument names / defaults of omniglot/model.py:15-28 and mimgnet/model.py:12-25."""
    from cmvae_amd.model import OmniglotGMVAE, FeatureGMVAE
    sig = inspect.signature(OmniglotGMVAE.__init__).parameters
    assert list(sig)[1:] == ['input_shape', 'unsupervised_em_iters', 'semisupervised_em_iters', 'fix_pi',
                             'hidden_size', 'component_size', 'latent_size', 'train_mc_sample_size',
                             'test_mc_sample_size', 'linear', 'eval_iter', 'laplace', 'gamma']
    assert sig['linear'].default is False and sig['component_size'].default == 20
    sigf = inspect.signature(FeatureGMVAE.__init__).parameters
    assert list(sigf)[1:] == ['input_shape', 'unsupervised_em_iters', 'semisupervised_em_iters', 'fix_pi',
                              'component_size', 'latent_size', 'train_mc_sample_size', 'test_mc_sample_size',
                              'linear', 'laplace', 'gamma']
    assert sigf['linear'].default is True and sigf['input_shape'].default == 512


def test_state_dict_keys_and_param_count():
    from cmvae_amd.model import OmniglotGMVAE, FeatureGMVAE
    m = OmniglotGMVAE()
    keys = set(m.state_dict())
    for k in ('encoder.encoder.0.weight', 'encoder.encoder.13.running_var', 'q_z_given_H.0.mab.fc_q.weight',
              'q_z_given_H.2.bias', 'causal.fc1.weight', 'causal.fc2.0.weight', 'causal.fc2.0.bias',
              'proj.4.weight', 'decoder.decoder.9.bias', 'decoder.decoder.7.num_batches_tracked',
              'log_norm_constant', 'log_2', 'uniform_pi'):
        assert k in keys, k
    assert sum(p.numel() for p in m.parameters()) == 1140545
    assert m.state_dict()['log_norm_constant'].dtype == torch.float64
    f = FeatureGMVAE(input_shape=256)
    assert 'causal.A' in f.state_dict() and 'log_2' not in f.state_dict()
    assert tuple(f.state_dict()['causal.A'].shape) == (64, 64)


def test_size_queries_survive_degenerate_shapes():
    """Host-only size queries with zero-sized shapes.  A 5-way episode with sample-size 5 has te = 0 query items, so
    Nu = 0 reached the tile partition in round 1 and died with SIGFPE (integer division by zero in the host code);
    the queries must return a size, the entry points reject the shape (tests/test_gpu_kernels.py)."""
    from cmvae_amd._lib import lib
    assert lib.cmvae_unsup_prior_workspace_bytes(2, 0, 5) > 0
    assert lib.cmvae_semi_prior_workspace_bytes(2, 0, 5, 5) > 0
    assert lib.cmvae_semi_prior_workspace_bytes(2, 10, 0, 5) > 0
    assert lib.cmvae_semi_prior_workspace_bytes(0, 0, 0, 5) >= 0
    assert lib.cmvae_mixture_workspace_bytes(2, 0, 5) >= 0
    for q in (lib.cmvae_conv_wgrad_workspace_bytes(0), lib.cmvae_linear_bwd_workspace_bytes(0, 64, 64),
              lib.cmvae_conv_gemm_grid(0, 1), lib.cmvae_rowpass_grid(0), lib.cmvae_dec_out_grid(0),
              lib.cmvae_causal_lc_bwd_grid(0), lib.cmvae_center_rows_chunks(0)):
        assert q >= 0


def test_torch_ops_namespace_lists_every_leaf_op():
    """torch.ops.cmvae_amd.* (cmvae_amd/ops.py + registry.py) exists for every registered name without a GPU; the
    geometry struct survives the trip through its int-list form."""
    import torch
    import cmvae_amd.registry as reg
    from cmvae_amd import conv
    for name in reg.REGISTERED:
        assert hasattr(torch.ops.cmvae_amd, name), name
    for g in (conv.geom_conv3x3(3, 7, 7), conv.geom_convT_phase(64, 7, 7, 4, 1, 0, pixel_major=True)[0],
              conv.geom_linear(10, 128, 256)):
        lst = reg.geom_to_list(g)
        assert reg.geom_to_list(reg.geom_from_list(lst)) == lst


def test_bf16_pipe_rules_restated_for_measurement_code():
    """`conv.on_bf16_pipe` / `conv.bf16_pipe_passes` restate the launch rules of csrc/conv_gemm.hip for bench.py's
    roofline line (host arithmetic only: no launch).  A 4 x 4 stride-2 data gradient has four tap parity classes of
    four taps = four accumulating passes, a 3 x 3 one 4 + 2 + 2 + 1 = two passes (6 + 3), a 3 x 3 conv at stride 1 two
    passes of tap order, at most six taps one launch."""
    from cmvae_amd import conv
    thr = conv.bf16_pipe_min_rows()
    assert thr == 3 * 256 * 256
    n = 1024
    g4, _ = conv.geom_convT_dgrad(n, 14, 14, 4, pixel_major=True)
    assert g4.ntaps == 16 and g4.rows_total >= thr
    assert conv.on_bf16_pipe('bnbwd', [g4]) and conv.bf16_pipe_passes('bnbwd', g4) == 4
    g3, _ = conv.geom_convT_dgrad(4 * n, 7, 7, 3, pixel_major=True)
    assert g3.ntaps == 9 and conv.on_bf16_pipe('bnbwd', [g3]) and conv.bf16_pipe_passes('bnbwd', g3) == 2
    gc = conv.geom_conv3x3(1280, 14, 14)
    assert gc.ntaps == 9 and conv.on_bf16_pipe('gemm', [gc]) and conv.bf16_pipe_passes('gemm', gc) == 2
    small, _ = conv.geom_convT_dgrad(64, 14, 14, 4, pixel_major=True)
    assert not conv.on_bf16_pipe('bnbwd', [small])
    ph, _ = conv.geom_convT_phase(n, 14, 14, 4, 0, 1, pixel_major=True)
    assert ph.ntaps <= 6 and conv.bf16_pipe_passes('bnbwd', ph) == 1
