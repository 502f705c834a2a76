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
import torch.nn.functional as F

from . import dist as _dist


class EpisodeBatchNorm2d(nn.BatchNorm2d):
    """BatchNorm2d whose training statistics span the whole meta-batch, across ranks.

    The reference normalises over all B*S (encoder) / B*S*M (decoder) images of the step
    (omniglot/module.py:15-31, 51-59).  When the episodes are sharded over GPUs the per-channel
    sums are all-reduced (collective C2) so the numbers do not depend on the
    number of ranks.  Single-rank behaviour equals nn.BatchNorm2d.
    """

    def forward(self, x):
        if self.training:
            return _dist.sync_batch_norm(x, self)
        C = x.shape[1]
        scale = self.weight * torch.rsqrt(self.running_var + self.eps)
        shift = self.bias - self.running_mean * scale
        return x * scale.view(1, C, 1, 1) + shift.view(1, C, 1, 1)


class GemmConv2d(nn.Conv2d):
    """3x3 / stride 1 / pad 1 convolution as im2col + GEMM (no JIT-compiled library kernels)."""

    def forward(self, x):
        n, _, h, w = x.shape
        cols = F.unfold(x, 3, padding=1)                               # [n, cin*9, h*w]
        out = self.weight.view(self.out_channels, -1).matmul(cols)      # [n, cout, h*w]
        return out.view(n, self.out_channels, h, w) + self.bias.view(1, -1, 1, 1)


class GemmConvTranspose2d(nn.ConvTranspose2d):
    """k x k / stride 2 / pad 1 transposed convolution as GEMM + col2im."""

    def forward(self, x):
        n, cin, h, w = x.shape
        k = self.kernel_size[0]
        ho, wo = (h - 1) * 2 - 2 + k, (w - 1) * 2 - 2 + k
        cols = self.weight.view(cin, -1).t().matmul(x.reshape(n, cin, h * w))   # [n, cout*k*k, h*w]
        out = F.fold(cols, (ho, wo), k, padding=1, stride=2)
        return out + self.bias.view(1, -1, 1, 1)


def _conv_block(cin, cout):
    return [GemmConv2d(cin, cout, 3, 1, 1), EpisodeBatchNorm2d(cout), nn.ReLU(inplace=True),
            nn.MaxPool2d(2, ceil_mode=True)]


class OmniglotEncoder(nn.Module):
    """28x28 -> 14 -> 7 -> 4 -> 2, flattened to 4*hidden features."""

    def __init__(self, hidden_size):
        super().__init__()
        self.last_hidden_size = 4 * hidden_size
        layers = _conv_block(1, hidden_size)
        for _ in range(3):
            layers += _conv_block(hidden_size, hidden_size)
        self.encoder = nn.Sequential(*layers)

    def forward(self, x):
        return self.encoder(x).reshape(-1, self.last_hidden_size)


class OmniglotDecoder(nn.Module):
    """[*, 4*hidden] -> view [*, hidden, 2, 2] -> 4 -> 7 -> 14 -> 28 (sigmoid)."""

    def __init__(self, hidden_size):
        super().__init__()
        self.hidden_size = hidden_size
        h = hidden_size
        self.decoder = nn.Sequential(
            GemmConvTranspose2d(h, h, 4, 2, 1), EpisodeBatchNorm2d(h), nn.ReLU(inplace=True),
            GemmConvTranspose2d(h, h, 3, 2, 1), EpisodeBatchNorm2d(h), nn.ReLU(inplace=True),
            GemmConvTranspose2d(h, h, 4, 2, 1), EpisodeBatchNorm2d(h), nn.ReLU(inplace=True),
            GemmConvTranspose2d(h, 1, 4, 2, 1), nn.Sigmoid())

    def forward(self, h):
        return self.decoder(h.view(-1, self.hidden_size, 2, 2))


class MAB(nn.Module):
    """Multi-head attention block of the Set Transformer, ln=False variant only."""

    def __init__(self, dim_Q, dim_K, dim_V, num_heads, ln=False):
        super().__init__()
        if ln:
            raise NotImplementedError('the CMVAE hot path never enables layer norm')
        self.dim_V, self.num_heads = dim_V, num_heads
        self.fc_q = nn.Linear(dim_Q, dim_V)
        self.fc_k = nn.Linear(dim_K, dim_V)
        self.fc_v = nn.Linear(dim_K, dim_V)
        self.fc_o = nn.Linear(dim_V, dim_V)

    def forward(self, Q, K):
        B, S, _ = Q.shape
        hd = self.dim_V // self.num_heads
        q = self.fc_q(Q)
        k = self.fc_k(K)
        v = self.fc_v(K)
        # heads as a batch axis; same arithmetic as the reference's split/cat-on-batch form
        qh = q.view(B, S, self.num_heads, hd).transpose(1, 2)
        kh = k.view(B, -1, self.num_heads, hd).transpose(1, 2)
        vh = v.view(B, -1, self.num_heads, hd).transpose(1, 2)
        att = torch.softmax(qh.matmul(kh.transpose(-1, -2)) / math.sqrt(self.dim_V), -1)
        o = (qh + att.matmul(vh)).transpose(1, 2).reshape(B, S, self.dim_V)
        return o + F.relu(self.fc_o(o))


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

    def forward(self, x):                      # x [n,d,m1]
        out = torch.einsum('ndm,dmo->ndo', x, self.weight)
        return out if self.bias is None else out + self.bias


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

    def mask(self, x):
        if self.linear:
            return x.matmul(self.A)
        h = self.fc1(x).view(-1, self.dims[0], self.dims[1])
        for fc in self.fc2:
            h = fc(F.elu(h))
        return h.squeeze(2).view(x.shape)

    def forward(self, x):
        return x - self.mask(x)

    def get_w(self):
        if self.linear:
            return self.A
        w = self.fc1.weight.view(self.latent_size, -1, self.latent_size)
        return w.pow(2).sum(1).t()
