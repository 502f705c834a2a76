"""Parameter containers of the CMVAE hot path, laid out so that ``state_dict()`` keys equal the
reference's: ``encoder.encoder.{0,1,4,5,8,9,12,13}.*``,
``q_z_given_H.{0,1}.mab.fc_{q,k,v,o}.*``, ``q_z_given_H.2.*``, ``causal.fc1.*``,
``causal.fc2.0.{weight,bias}`` / ``causal.A``, ``proj.{0,2,4}.*``,
``decoder.decoder.{0,1,3,4,6,7,9}.*``.

Mirrors (behaviour, not code): omniglot/module.py:7-40 (encoder), 43-71 (decoder), 75-112
(MAB/SAB), 136-196 (CausalEncoder), 232-288 (LocallyConnected); mimgnet/module.py:47-110.

Unlike the reference nothing here calls ``.cuda()``: modules are built on the CPU and moved
with ``.to(device)``.
"""
import math

import torch
import torch.nn as nn

from . import dist as _dist
from .linear import linear as _linear, causal_mlp as _causal_mlp, attention as _attention, \
    self_attention as _self_attention

class EpisodeBatchNorm2d(nn.BatchNorm2d):
    """Parameters and running statistics of a BatchNorm2d whose training statistics span the whole
    meta-batch, across ranks (omniglot/module.py:15-31, 51-59; SURVEY.md §5, collective C2).  The
    arithmetic lives in the HIP pipelines (cmvae_amd/encoder.py, decoder.py: GEMM-epilogue sums ->
    all-reduce -> cmvae_bn_finalize); this class only holds the state under the reference's keys."""

    def forward(self, x):
        raise RuntimeError('EpisodeBatchNorm2d is a parameter container: run the encoder / decoder through '
                           'cmvae_amd.encoder.encode_images / cmvae_amd.decoder.decoder_bce')


def _conv_block(cin, cout):
    return [nn.Conv2d(cin, cout, 3, 1, 1), EpisodeBatchNorm2d(cout), nn.ReLU(inplace=True),
            nn.MaxPool2d(2, ceil_mode=True)]


class OmniglotEncoder(nn.Module):
    """28x28 -> 14 -> 7 -> 4 -> 2, flattened to 4*hidden features (parameter container; the HIP
    pipeline is cmvae_amd.encoder.encode_images)."""

    def __init__(self, hidden_size):
        super().__init__()
        self.last_hidden_size = 4 * hidden_size
        layers = _conv_block(1, hidden_size)
        for _ in range(3):
            layers += _conv_block(hidden_size, hidden_size)
        self.encoder = nn.Sequential(*layers)

    def forward(self, x):
        """x [n,1,28,28] (or [n,784]) -> [n, 4*hidden] in the reference's c*4+pix column order."""
        from .encoder import encode_images
        from .decoder import nhwc_perm, permute
        H = encode_images(self, x.reshape(x.shape[0], -1))
        perm, inv = nhwc_perm(H.device, self.last_hidden_size // 4)
        return permute(H, 1, inv, perm)          # reference column c*4 + pix takes channels-last column pix*hs + c


class OmniglotDecoder(nn.Module):
    """[*, 4*hidden] -> view [*, hidden, 2, 2] -> 4 -> 7 -> 14 -> 28 (sigmoid).  Parameter container; the
    HIP pipeline (fused with the Bernoulli likelihood) is cmvae_amd.decoder.decoder_bce."""

    def __init__(self, hidden_size):
        super().__init__()
        self.hidden_size = hidden_size
        h = hidden_size
        self.decoder = nn.Sequential(
            nn.ConvTranspose2d(h, h, 4, 2, 1), EpisodeBatchNorm2d(h), nn.ReLU(inplace=True),
            nn.ConvTranspose2d(h, h, 3, 2, 1), EpisodeBatchNorm2d(h), nn.ReLU(inplace=True),
            nn.ConvTranspose2d(h, h, 4, 2, 1), EpisodeBatchNorm2d(h), nn.ReLU(inplace=True),
            nn.ConvTranspose2d(h, 1, 4, 2, 1), nn.Sigmoid())

    def forward(self, h):
        raise RuntimeError('OmniglotDecoder is a parameter container: use cmvae_amd.decoder.decoder_bce')


class MAB(nn.Module):
    """Multi-head attention block of the Set Transformer, ln=False variant only."""

    def __init__(self, dim_Q, dim_K, dim_V, num_heads, ln=False):
        super().__init__()
        if ln:
            raise NotImplementedError('the CMVAE hot path never enables layer norm')
        if dim_V != 64 * num_heads or dim_Q != dim_K:
            raise ValueError('the MI355X attention kernel needs heads of 64 channels and self-attention shapes')
        self.dim_V, self.num_heads = dim_V, num_heads
        self.fc_q = nn.Linear(dim_Q, dim_V)
        self.fc_k = nn.Linear(dim_K, dim_V)
        self.fc_v = nn.Linear(dim_K, dim_V)
        self.fc_o = nn.Linear(dim_V, dim_V)

    def forward(self, Q, K):
        if Q is K and Q.dim() == 3 and Q.shape[0] * Q.shape[1] > 0:
            # self-attention: the three projections read the same rows -> one GEMM over the stacked weights
            o = _self_attention(Q, self.fc_q.weight, self.fc_q.bias, self.fc_k.weight, self.fc_k.bias,
                                self.fc_v.weight, self.fc_v.bias, self.num_heads, 1.0 / math.sqrt(self.dim_V))
            return o + _linear(o, self.fc_o.weight, self.fc_o.bias, 'relu')
        q = _linear(Q, self.fc_q.weight, self.fc_q.bias)
        k = _linear(K, self.fc_k.weight, self.fc_k.bias)
        v = _linear(K, self.fc_v.weight, self.fc_v.bias)
        # heads are 64-column slices of the rows; softmax scale is 1/sqrt(dim_V) as in the reference
        o = _attention(q, k, v, self.num_heads, 1.0 / math.sqrt(self.dim_V))
        return o + _linear(o, self.fc_o.weight, self.fc_o.bias, 'relu')


class SAB(nn.Module):
    def __init__(self, dim_in, dim_out, num_heads, ln=False):
        super().__init__()
        self.mab = MAB(dim_in, dim_in, dim_out, num_heads, ln=ln)

    def forward(self, X):
        return self.mab(X, X)


class LocallyConnected(nn.Module):
    """d independent m1->m2 linear maps: weight [d,m1,m2], bias [d,m2]; U(+-1/sqrt(m1)) init."""

    def __init__(self, num_linear, input_features, output_features, bias=True):
        super().__init__()
        self.num_linear, self.input_features, self.output_features = num_linear, input_features, output_features
        self.weight = nn.Parameter(torch.empty(num_linear, input_features, output_features))
        if bias:
            self.bias = nn.Parameter(torch.empty(num_linear, output_features))
        else:
            self.register_parameter('bias', None)
        bound = math.sqrt(1.0 / input_features)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def forward(self, x):
        raise RuntimeError('LocallyConnected is a parameter container: CausalEncoder.forward runs the ELU + locally '
                           'connected tail in cmvae_causal_lc_fwd / cmvae_causal_rows')


class CausalEncoder(nn.Module):
    """z -> eps = z - f(z): linear adjacency (f(z) = z A) or NOTEARS-MLP (fc1 64->64*32, ELU,
    locally connected 32->1)."""

    def __init__(self, latent_size, bias=True, linear=True):
        super().__init__()
        self.latent_size, self.linear = latent_size, linear
        if linear:
            self.A = nn.Parameter(torch.zeros(latent_size, latent_size))
        else:
            self.dims = [latent_size, 32, 1]
            self.fc1 = nn.Linear(latent_size, latent_size * self.dims[1], bias=bias)
            self.fc2 = nn.ModuleList([LocallyConnected(latent_size, self.dims[1], self.dims[2], bias=bias)])
            nn.init.xavier_normal_(self.fc1.weight.data)       # bias keeps nn.Linear's default init

    def forward(self, x):
        """eps = x - mask(x) on the HIP kernels (bulk rows; the EM loop uses cmvae_causal_rows instead)."""
        if self.linear:
            return x - _linear(x, self.A.t().contiguous())
        return _causal_mlp(x, self.fc1.weight, self.fc1.bias, self.fc2[0].weight, self.fc2[0].bias)

    def get_w(self):
        if self.linear:
            return self.A
        w = self.fc1.weight.view(self.latent_size, -1, self.latent_size)
        return w.pow(2).sum(1).t()
