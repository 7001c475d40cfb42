"""Oracle temporal-graph augmentation block.  Test infrastructure -- see oracle/__init__.py.

Follows lib/ops/module_wrappers/temporal_graph.py.  The reference block cannot be
constructed as shipped (ctor calls ``reset_parameter``, the method is
``reset_parameters``, :117/:124 vs :131) and ``build_aug_block`` (lib/ops/build.py:9-32)
raises for every name; the *math* of ``forward`` (:227-239) is the spec and is what is
restated here.  RelaxedBernoulli sampling (:187-192) is made injectable (uniform noise
``u``) so parity tests are deterministic.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def hop_distance(T, max_hop=3):
    """temporal_graph.py:7-36: chain graph over T frames (self + i<->i+1 links);
    hop_dis[i,j] = smallest d<=max_hop with (A^d)[i,j]>0, else inf.  For a chain that is
    |i-j| when <= max_hop."""
    idx = torch.arange(T)
    d = (idx[:, None] - idx[None, :]).abs().float()
    return torch.where(d <= max_hop, d, torch.full_like(d, float('inf')))


def theta(hop, alpha=0.5):
    """temporal_graph.py:206: exp(-h) / (1 + exp(-h)**2) + alpha."""
    return math.exp(-hop) / (1 + math.exp(-hop) ** 2) + alpha


def relaxed_bernoulli_rsample(probs, u, temperature=1.0):
    """torch.distributions.RelaxedBernoulli(temperature, probs).rsample() with the uniform
    noise made explicit (LogitRelaxedBernoulli.rsample: probs clamped to [eps, 1-eps] by
    probs_to_logits, u clamped by clamp_probs; sigmoid((logit(u) + logit(p)) / T))."""
    eps = torch.finfo(probs.dtype).eps
    p = probs.clamp(min=eps, max=1 - eps)
    logits = torch.log(p) - torch.log1p(-p)
    uu = u.clamp(min=eps, max=1 - eps)
    return torch.sigmoid((uu.log() - (-uu).log1p() + logits) / temperature)


class GCNLayer(nn.Module):
    """temporal_graph.py:38-64: support = conv1x1x1(x); out = einsum('bij,bcjhw->bcihw') + support."""

    def __init__(self, cin, cout=None, bias=False):
        super().__init__()
        self.conv = nn.Conv3d(cin, cin if cout is None else cout, kernel_size=(1, 1, 1), bias=bias)

    def forward(self, x, adj):
        s = self.conv(x)
        return torch.einsum('bij,bcjhw->bcihw', adj, s) + s


class TemporalGraphAug(nn.Module):
    """temporal_graph.py:66-239 with the defaults of :67-70 (sub_sample, max_pool, no bias,
    no bn_layer, one GCN layer C->C because `inter_channels` (None) is what :95 passes)."""

    def __init__(self, in_channels, alpha=0.5, temperature=1.0, max_hop=3):
        super().__init__()
        self.in_channels, self.alpha, self.temperature, self.max_hop = in_channels, alpha, temperature, max_hop
        inter = max(in_channels // 2, 1)
        self.gcns = nn.ModuleList([GCNLayer(in_channels)])
        pool = nn.MaxPool3d(kernel_size=(1, 2, 2))
        q = nn.Conv3d(in_channels, inter, 1, bias=False)
        k = nn.Conv3d(in_channels, inter, 1, bias=False)
        std = 1.0 / math.sqrt(in_channels)                      # reset_parameters :131-148
        q.weight.data.uniform_(-std, std)
        k.weight.data.uniform_(-std, std)
        self.g_q = nn.Sequential(q, pool)
        self.g_k = nn.Sequential(k, pool)

    def sim_adj(self, x):
        """:150-178 -- softmax_j( <g_q(x)_i , g_k(x)_j> ) over frames."""
        B, T = x.size(0), x.size(2)
        gq = self.g_q(x).transpose(2, 1).contiguous().view(B, T, -1)
        gk = self.g_k(x).transpose(2, 1).contiguous().view(B, T, -1)
        return F.softmax(torch.matmul(gq, gk.permute(0, 2, 1)), dim=-1)

    def hop_weighted(self, sim, hop):
        """:204-210."""
        adj = torch.zeros_like(sim)
        for h in range(self.max_hop + 1):
            m = hop == h
            adj[:, m] = sim[:, m] * theta(h, self.alpha)
        return adj

    def forward(self, x, u=None, adj=None):
        """:227-239.  `u` = uniform noise (B,T,T) for the RelaxedBernoulli rsample; `adj`
        short-circuits everything up to and including the sampling."""
        if adj is None:
            hop = hop_distance(x.size(2), self.max_hop)
            pre = self.hop_weighted(self.sim_adj(x), hop)
            if u is None:
                u = torch.rand_like(pre)
            adj = relaxed_bernoulli_rsample(pre, u, self.temperature)
        for g in self.gcns:
            x = g(x, adj)
        return x


class AugThen(nn.Sequential):
    """Intended result of build_aug_block (lib/ops/build.py:20-21): Sequential(aug, module)."""

    def __init__(self, aug, module):
        super().__init__(aug, module)
        self.noise = None      # optional injected uniform noise for the aug block

    def forward(self, x):
        return self[1](self[0](x, u=self.noise))


def _resolve(root, dotted):
    parts = dotted.split('.')
    parent = root
    for p in parts[:-1]:
        parent = getattr(parent, p)
    return parent, parts[-1]


def first_conv_in_channels(module):
    for m in module.modules():
        if isinstance(m, nn.Conv3d):
            return m.in_channels
    raise ValueError('no Conv3d inside %r' % type(module))


def build_aug_block(base_model, module_name_list, n_segments=None):
    """Working equivalent of lib/ops/build.py:9-32: EVERY named sub-module m becomes
    Sequential(TemporalGraphAug(C_in(m)), m); C_in is the first conv's in_channels."""
    for name in module_name_list:
        parent, leaf = _resolve(base_model, name)
        mod = getattr(parent, leaf)
        setattr(parent, leaf, AugThen(TemporalGraphAug(first_conv_in_channels(mod)), mod))
    return base_model
