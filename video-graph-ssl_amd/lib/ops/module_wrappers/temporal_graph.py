"""Temporal-graph augmentation block on the HIP engine.

Reference: lib/ops/module_wrappers/temporal_graph.py.  The reference block cannot even be
constructed (ctor calls the non-existent ``reset_parameter``, :117/:124) and is never inserted
(SURVEY.md fact 5); what is implemented here is the arithmetic of its ``forward`` (:227-239) with the
defaults of :67-70: hop-distance graph over the T frames (:7-36), similarity adjacency from two
1x1x1 convs + (1,2,2) max-pool + T x T matmul + softmax (:150-178), hop weighting theta(h) (:204-210),
RelaxedBernoulli re-sampling (:187-192) and ONE GCN layer C->C: 1x1x1 conv, dense neighbourhood
aggregation einsum('bij,bcjhw->bcihw') and skip (:56-64).  Parameter names match the reference
(gcns.0.conv.weight, g_q.0.weight, g_k.0.weight).
"""
import math

import torch
import torch.nn as nn

from ....engine import layers as L, ops
from ....engine.layers import HipConv3d, HipMaxPool3d
from ....engine.tape import Var


class TemporalGraph(object):
    """Hop distances of the chain graph over `tem_len` frames: |i-j| if <= max_hop else inf (:7-36)."""

    def __init__(self, tem_len=16, max_hop=1, dilation=1):
        self.max_hop, self.dilation, self.num_node = max_hop, dilation, tem_len
        idx = torch.arange(tem_len)
        d = (idx[:, None] - idx[None, :]).abs().float()
        self.temporal_graph = torch.where(d <= max_hop, d, torch.full_like(d, float('inf')))


class GCN(nn.Module):
    def __init__(self, in_features, out_features=None, bias=False, skip=True):
        super().__init__()
        if not skip:
            raise NotImplementedError('the reference always uses skip=True')
        self.skip = skip
        self.in_features = in_features
        self.out_features = in_features if out_features is None else out_features
        self.conv = HipConv3d(in_features, self.out_features, (1, 1, 1), bias=bias)

    def fwd(self, tape, xv, adjv):
        sv = L.f_conv(tape, self.conv, xv)
        s, adj = sv.t, adjv.t
        out = ops.graph_gcn_fwd(adj, s)
        ov = Var(out, tape.recording)

        def back():
            ds, dadj = ops.graph_gcn_bwd(adj, s, ov.grad, want_dadj=adjv.needs_grad)
            ov.grad = None
            sv.add_grad(ds)
            if adjv.needs_grad:
                adjv.add_grad(dadj)
        tape.record(back)
        return ov


class _ConvBN(nn.Sequential):
    """Sequential(Conv3d, BatchNorm3d): the similarity branch under bn_layer=True (temporal_graph.py:103-113)."""

    def fwd(self, tape, xv):
        return L.f_conv_bn_act(tape, self[0], self[1], xv, relu=False)


class TemporalGraphAug(nn.Module):
    """Constructor options as in the reference (:66-129).  The shipped configs use the defaults; the others follow the
    reference's module layout so its state-dict keys load: `sub_sample=False` leaves g_q / g_k as bare convs (keys
    g_q.weight), `max_pool=False` pools with AvgPool3d((1,2,2)), `bn_layer=True` puts a BatchNorm3d behind each similarity
    conv (keys g_q.0.0.weight, g_q.0.1.*), `num_gcn_layers > 1` stacks GCN(C->inter), GCN(inter->inter)..., GCN(inter->C)
    and needs an explicit `inter_channels` (with inter_channels=None the reference passes None as in_features and cannot be
    built, :95-98).  `mask_frame` raises: the reference's mask indexes the BATCH axis with the frame index (:171-174)."""

    def __init__(self, in_channels, inter_channels=None, sub_sample=True, bias=False, bn_layer=False,
                 zero_init=False, max_pool=True, mask_frame=False, nei_size=None, alpah=0.5, num_gcn_layers=1,
                 temperature=1., max_hop=3):
        super().__init__()
        if mask_frame:
            raise NotImplementedError('mask_frame: the reference masks adjacent_matrix[i] by FRAME index i on the batch '
                                      'axis (temporal_graph.py:171-174); there is no well-defined arithmetic to match')
        if num_gcn_layers < 1 or (num_gcn_layers > 1 and inter_channels is None):
            raise ValueError('num_gcn_layers > 1 needs inter_channels (the reference builds GCN(in_features=None) otherwise, '
                             'temporal_graph.py:95-98)')
        self.in_channels = in_channels
        self.sub_sample, self.bn_layer, self.max_pool, self.zero_init = sub_sample, bn_layer, max_pool, zero_init
        self.num_gcn_layers = num_gcn_layers
        gcn_mid = inter_channels                                # what :95 passes as out_features (None -> C)
        self.inter_channels = max(in_channels // 2, 1) if inter_channels is None else inter_channels
        self.alpha, self.temperature, self.max_hop, self.bias = alpah, temperature, max_hop, bias
        self.gcns = nn.ModuleList([GCN(in_features=in_channels, out_features=gcn_mid)])
        for i in range(1, num_gcn_layers):
            self.gcns.append(GCN(in_features=gcn_mid, out_features=in_channels) if i == num_gcn_layers - 1
                             else GCN(in_features=gcn_mid))
        q = HipConv3d(in_channels, self.inter_channels, 1, 1, 0, bias=bias)
        k = HipConv3d(in_channels, self.inter_channels, 1, 1, 0, bias=bias)
        self.reset_parameters(q, k, zero_init)
        if bn_layer:
            q, k = _ConvBN(q, L.HipBatchNorm3d(self.inter_channels)), _ConvBN(k, L.HipBatchNorm3d(self.inter_channels))
        if sub_sample:
            pool = HipMaxPool3d(kernel_size=(1, 2, 2)) if max_pool else L.HipAvgPool3d(kernel_size=(1, 2, 2))
            q, k = nn.Sequential(q, pool), nn.Sequential(k, pool)
        self.g_q, self.g_k = q, k
        self.noise = None          # optional injected uniforms (B,T,T) for deterministic parity tests

    def reset_parameters(self, m1, m2, zero_init=False):
        """Uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) for the two similarity convs (:131-148)."""
        for m in (m1, m2):
            if zero_init:
                nn.init.constant_(m.weight, 0)
            else:
                n = m.kernel_size[0] * m.kernel_size[1] * m.kernel_size[2] * m.in_channels
                m.weight.data.uniform_(-1. / math.sqrt(n), 1. / math.sqrt(n))
            if m.bias is not None:
                nn.init.constant_(m.bias, 0) if zero_init else m.bias.data.uniform_(-1. / math.sqrt(n), 1. / math.sqrt(n))

    def _branch(self, tape, mod, xv):
        if isinstance(mod, HipConv3d):
            return L.f_conv(tape, mod, xv)
        return L.f_seq(tape, mod, xv)            # Sequential of conv | _ConvBN, then the pool

    def fwd(self, tape, xv):
        x = xv.t
        B, T = x.shape[0], x.shape[2]
        gq = self._branch(tape, self.g_q, xv)
        gk = self._branch(tape, self.g_k, xv)
        u = self.noise
        if u is None:
            u = torch.rand((B, T, T), device=x.device)        # RelaxedBernoulli.rsample's uniforms
        sim, pre, adj = ops.graph_adj_fwd(gq.t, gk.t, u, self.max_hop, self.alpha, self.temperature)
        adjv = Var(adj, tape.recording)

        def back():
            if adjv.grad is not None:
                dgq, dgk = ops.graph_adj_bwd(adjv.grad, gq.t, gk.t, sim, pre, adj, self.max_hop, self.alpha,
                                             self.temperature)
                adjv.grad = None
                gq.add_grad(dgq)
                gk.add_grad(dgk)
        tape.record(back)
        out = xv
        for g in self.gcns:
            out = g.fwd(tape, out, adjv)
        return out


class AugThen(nn.Sequential):
    """Sequential(TemporalGraphAug, module): what build_aug_block installs in place of `module`."""

    def fwd(self, tape, xv):
        return L.f_seq(tape, self[1], self[0].fwd(tape, xv))
