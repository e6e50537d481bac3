"""Spatial graph-conv units behind the reference's class names and ctor kwargs.

``dgphgcn1`` (reference: pyskl/models/gcns/utils/gcn.py:2074-2372) is the DS-GCN dynamic-semantic
unit.  The modules own the same parameters/buffers under the same state_dict keys (built in the
same order, so a given torch seed yields the reference's initial weights), but ``forward`` is a
short chain of fused HIP ops (``dsgcn_amd.kernels``) instead of ~330 ATen calls:

    xbar --K-B dynadj--> Ahat        x --K-C pwconv--> Zp (+batch stats)
    (Zp, BN affine, ReLU) x Ahat --K-A aggregate--> Y --K-C pwconv--> Zo (+stats)

Train-mode BatchNorm never runs as its own pass: the producing conv emits the batch statistics,
the consumer applies ``scale*z+shift`` (+ReLU, +residual) while loading (a ``Deferred`` value).
"""
from math import ceil
from typing import NamedTuple, Optional, Tuple

import torch
import torch.nn as nn

import contextlib

from . import kernels


class Deferred(NamedTuple):
    """relu?( x1*a1.scale + a1.shift [+ x2*a2.scale + a2.shift | + x2] ) — not materialised."""
    x1: torch.Tensor
    a1: Optional[Tuple[torch.Tensor, torch.Tensor]]
    x2: Optional[torch.Tensor]
    a2: Optional[Tuple[torch.Tensor, torch.Tensor]]
    relu: bool

    def materialize(self):
        return kernels.ops().fuse_out(self.x1, self.a1, self.x2, self.a2, self.relu, False)[0]


def as_deferred(x):
    return x if isinstance(x, Deferred) else Deferred(x, None, None, None, False)


_pending_running = []     # (bn, mean, var, count) recorded during a training forward, applied by flush_running_stats


def record_running(bn, mean, var, count):
    if bn.training and bn.track_running_stats and mean is not None:
        _pending_running.append((bn, mean.detach(), var.detach(), float(count)))


@torch.no_grad()
def flush_running_stats():
    """Apply the running-statistics updates of every BatchNorm touched since the last flush, the way
    ``F.batch_norm(training=True)`` does (momentum, unbiased variance, num_batches_tracked): ONE launch for all layers
    (``kernels.bn_running_update``; three tiny kernels per layer in the reference)."""
    if not _pending_running:
        return
    items = list(_pending_running)
    _pending_running.clear()
    if kernels.FUSED_ENDS and all(bn.momentum is not None for bn, _, _, _ in items):
        kernels.ops().bn_running_update(items)
        return
    # the multi-tensor form (A/B switch; momentum=None = cumulative average, which needs the count on the host)
    torch._foreach_add_([bn.num_batches_tracked for bn, _, _, _ in items], 1)
    groups = {}
    for it in items:
        bn = it[0]
        m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
        groups.setdefault(float(m), []).append(it)
    for m, its in groups.items():
        rm = [bn.running_mean for bn, _, _, _ in its]
        rv = [bn.running_var for bn, _, _, _ in its]
        unb = torch._foreach_mul([var for _, _, var, _ in its], [c / max(c - 1.0, 1.0) for _, _, _, c in its])
        torch._foreach_mul_(rm, 1.0 - m)
        torch._foreach_add_(rm, [mean for _, mean, _, _ in its], alpha=m)
        torch._foreach_mul_(rv, 1.0 - m)
        torch._foreach_add_(rv, unb, alpha=m)


def _need_stats(bn):
    return bn.training or not bn.track_running_stats


def eval_affine(bn):
    scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
    return scale, bn.bias - bn.running_mean * scale


def op_bn(bn, op, count):
    """Run ``op(gamma, beta, eps, want_bn) -> (out, scale, shift, mean, var)`` with BatchNorm ``bn`` fused as a deferred
    affine: batch statistics in training (running stats recorded), running statistics in eval.  -> (out, (scale, shift))"""
    if _need_stats(bn):
        out, sc, sh, mean, var = op(bn.weight, bn.bias, bn.eps, True)
        record_running(bn, mean, var, count(out))
        return out, (sc, sh)
    return op(None, None, bn.eps, False)[0], eval_affine(bn)


def conv_bn(x1, a1, x2, a2, relu, conv, stride, aug, bn):
    """1x1 conv (module ``conv``) of a virtual input followed by BatchNorm ``bn`` as a deferred affine.
    -> (z, zaug, (scale, shift))"""
    ops = kernels.ops()
    n, _, T, V = x1.shape
    if _need_stats(bn):
        z, zaug, sc, sh, mean, var = ops.pwconv(x1, a1, x2, a2, relu, conv.weight, conv.bias, stride, aug,
                                                bn.weight, bn.bias, bn.eps, bn.num_features, True)
        record_running(bn, mean, var, n * z.shape[2] * (V + (1 if aug else 0)))
        return z, zaug, (sc, sh)
    z, zaug = ops.pwconv(x1, a1, x2, a2, relu, conv.weight, conv.bias, stride, aug)[:2]
    return z, zaug, eval_affine(bn)


def _norm_layer(norm, num_features):
    cfg = dict(norm) if isinstance(norm, dict) else dict(type=norm)
    typ = cfg.pop('type')
    if typ not in ('BN', 'BN2d'):
        raise NotImplementedError(f'norm type {typ} is not on the DS-GCN hot path')
    cfg.setdefault('eps', 1e-5)
    return nn.BatchNorm2d(num_features, **cfg)


def _check_act(act):
    typ = act['type'] if isinstance(act, dict) else act
    if typ != 'ReLU':
        raise NotImplementedError(f'activation {typ} is not fused by the HIP kernels (ReLU only)')


class dgphgcn1(nn.Module):
    """Dynamic-semantic graph conv.  Implemented configuration = the one every shipped DS-STGCN
    config selects (configs/dsstgcn/DSSTGCN_model.py:10-27): decompose + node/edge attention +
    subset-wise alpha/beta, ctr='T', ada='T', tanh / softmax.  Other flag combinations of the
    reference class are research variants outside BASELINE's configs and raise."""

    def __init__(self, in_channels, out_channels, A, edge_type, node_type, ratio=0.25, decompose=False, ctr='T',
                 ada='T', node_attention=False, edge_attention=False, ada_attention=False, target_specific=False,
                 add_type=False, sub_att=True, stage=True, num_types=5, edge_num=15, subset_wise=True,
                 ada_act='softmax', ctr_act='tanh', norm='BN', act='ReLU'):
        super().__init__()
        assert ada_act in ['tanh', 'relu', 'sigmoid', 'softmax']
        assert ctr_act in ['tanh', 'relu', 'sigmoid', 'softmax']
        assert ctr in [None, 'NA', 'T']
        assert ada in [None, 'NA', 'T']
        supported = (decompose and node_attention and edge_attention and subset_wise and sub_att and stage
                     and not ada_attention and not target_specific and not add_type and ctr == 'T' and ada == 'T'
                     and ada_act == 'softmax' and ctr_act == 'tanh')
        if not supported:
            raise NotImplementedError(
                'dgphgcn1: only the shipped DS-STGCN configuration (decompose, node_attention, edge_attention, '
                "subset_wise, sub_att, ctr='T', ada='T', tanh/softmax) has a HIP path")
        _check_act(act)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.num_subsets = K = A.size(0)
        self.num_types = num_types
        self.edge_num = edge_num
        if ratio is None:
            ratio = 1 / K
        self.ratio = ratio
        self.mid_channels = mid = int(ratio * out_channels)
        self.semantic_num = S = ceil(K / 3)
        self.norm_num = K - S
        if K != 3:
            raise NotImplementedError('the HIP dynadj kernel implements K=3 subsets (2 plain + 1 semantic)')
        self.register_buffer('node_type_idx', torch.as_tensor(node_type).to(torch.int32).contiguous(),
                             persistent=False)
        self.register_buffer('edge_type_idx', torch.as_tensor(edge_type).to(torch.int32).contiguous(),
                             persistent=False)
        # parameter creation order follows the reference ctor (same RNG consumption, same key order)
        self.A = nn.Parameter(A.clone())
        self.pre = nn.Sequential(nn.Conv2d(in_channels, mid * K, 1), _norm_layer(norm, mid * K), nn.ReLU())
        self.post = nn.Conv2d(mid * K, out_channels, 1)
        self.alpha = nn.Parameter(torch.zeros(K))
        self.beta = nn.Parameter(torch.zeros(K))
        self.conv1_se = nn.Conv2d(in_channels, S * mid * num_types, kernel_size=1)
        self.conv2_se = nn.Conv2d(in_channels, S * mid * num_types, kernel_size=1)   # never used (quirk Q1)
        self.conv1 = nn.Conv2d(in_channels, self.norm_num * mid, 1)
        self.conv2 = nn.Conv2d(in_channels, self.norm_num * mid, 1)
        self.edge_linears = nn.Conv2d(S * mid, edge_num * S * mid, 1)
        if in_channels != out_channels:
            self.down = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1), _norm_layer(norm, out_channels))
        else:
            self.down = None
        self.bn = _norm_layer(norm, out_channels)

    def flat_groups(self):
        """(see dgmstcn.flat_groups) the three mean-pooled projections run as one conv over their stacked weights"""
        return [[self.conv1.weight, self.conv2.weight, self.conv1_se.weight],
                [self.conv1.bias, self.conv2.bias, self.conv1_se.bias]]

    def fusable_pairs(self):
        """(conv, bn) pairs whose BatchNorm consumes the conv's output directly (checkpoint.fuse_conv_bn); `pre` and
        `down` are Sequential(conv, bn)."""
        return [(self.post, self.bn)]

    def adjacency(self, xbar, host=None):
        """Ahat (n, K*mid, V, V) from the time-averaged input xbar (n, Ci, V).  host: a kernels.bn_batch whose waiting
        finalize jobs K-B's launch carries."""
        c1, c2, cs, el = self.conv1, self.conv2, self.conv1_se, self.edge_linears
        kw = {} if host is None else dict(host=host)
        return kernels.ops().dynadj(
            xbar, self.A, self.alpha, self.beta,
            c1.weight.flatten(1), c1.bias, c2.weight.flatten(1), c2.bias, cs.weight.flatten(1), cs.bias,
            el.weight.flatten(1), el.bias, self.node_type_idx, self.edge_type_idx, **kw)

    def forward_deferred(self, x, xbar=None, x_res=None):
        """x_res: an alias of x for the residual operand (lets the caller route the gradients of the two uses of x
        separately, see kernels.tee3)."""
        ops = kernels.ops()
        x_res = x if x_res is None else x_res
        if xbar is None:                          # first block: joint rows padded to 32 like fuse_out(want_tmean=32)'s
            xbar = ops.tmean(x, 32) if x.shape[-1] <= 32 else ops.tmean(x)
        fork = getattr(ops, 'side_branch', None)
        batch = getattr(ops, 'bn_batch', None)
        if batch is not None and not getattr(ops, 'OVERLAP', False):
            # `pre` conv first, K-B behind it: K-B (and its projection conv) read nothing of the `pre` BatchNorm, so its
            # finalize rides in K-B's launch as extra workgroups (kernels.bn_batch) instead of a launch of its own between
            # the conv and K-A; the backward mirrors it (K-A's rows -> the coefficient job hosted by K-B's backward)
            with batch() as q:
                zp, _, ap = conv_bn(x, None, None, None, False, self.pre[0], 1, False, self.pre[1])
                ahat = self.adjacency(xbar, host=q)
            ctxbn = getattr(ap[0], '_dsgcn_bn', None)
            if ctxbn is not None:
                ctxbn.host = True
        elif fork is None:
            ahat = self.adjacency(xbar)
            zp, _, ap = conv_bn(x, None, None, None, False, self.pre[0], 1, False, self.pre[1])
        else:
            with fork(x) as br:                    # K-B beside the `pre` channel mix (independent until K-A)
                ahat = self.adjacency(xbar)
            zp, _, ap = conv_bn(x, None, None, None, False, self.pre[0], 1, False, self.pre[1])
            br.join(ahat)
        y = ops.aggregate(zp, ap, True, ahat)
        if self.down is None:
            zo, _, ao = conv_bn(y, None, None, None, False, self.post, 1, False, self.bn)
            return Deferred(zo, ao, x_res, None, True)
        # `post` and `down` are independent of one another: their two finalizes are one launch
        with (batch() if batch is not None else contextlib.nullcontext()):
            zo, _, ao = conv_bn(y, None, None, None, False, self.post, 1, False, self.bn)
            zd, _, ad = conv_bn(x_res, None, None, None, False, self.down[0], 1, False, self.down[1])
        return Deferred(zo, ao, zd, ad, True)

    def forward(self, x, A=None):
        out = self.forward_deferred(x).materialize()
        flush_running_stats()          # standalone use: running statistics move with the call, as in F.batch_norm
        return out

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out')
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


class dggcn(nn.Module):
    """The original DG-STGCN spatial unit (reference: pyskl/models/gcns/utils/gcn.py:1445-1584; selected by
    ``DGSTGCN(gcn_type='dggcn')``, dgstgcn.py:42-43): K subsets, each with its own mean-pooled projections x1_k = conv1_k
    (xbar), x2_k = conv2_k(xbar), ``Ahat_k = A_k + alpha * tanh(x1_k[u] - x2_k[w]) + beta * softmax_u(sum_c x1_k x2_k)`` — the
    ``dgphgcn1`` arithmetic without the node-typed select and the edge-typed linear.  Implemented configuration: the
    class defaults (ctr='T', ada='T', tanh / softmax), subset_wise on or off.

    K-B (dsgcn_dynadj_*) evaluates one plain subset pair (a0-b0, a1-b1) exactly when its edge-typed linear is the
    identity (one edge class, We = I, be = 0); the third subset goes through a second call whose first slot carries
    (a2, b2).  Two launches instead of one and a concatenation of the result: this unit is an f-4 row, not a bench path."""

    def __init__(self, in_channels, out_channels, A, ratio=0.25, ctr='T', ada='T', subset_wise=False, ada_act='softmax',
                 ctr_act='tanh', norm='BN', act='ReLU'):
        super().__init__()
        assert ada_act in ['tanh', 'relu', 'sigmoid', 'softmax']
        assert ctr_act in ['tanh', 'relu', 'sigmoid', 'softmax']
        assert ctr in [None, 'NA', 'T']
        assert ada in [None, 'NA', 'T']
        if not (ctr == 'T' and ada == 'T' and ada_act == 'softmax' and ctr_act == 'tanh'):
            raise NotImplementedError("dggcn: only ctr='T', ada='T', tanh / softmax (the class defaults) has a HIP path")
        _check_act(act)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.num_subsets = K = A.size(0)
        if K != 3:
            raise NotImplementedError('the HIP dynadj kernel implements K=3 subsets')
        self.subset_wise = subset_wise
        if ratio is None:
            ratio = 1 / K
        self.ratio = ratio
        self.mid_channels = mid = int(ratio * out_channels)
        # parameter creation order follows the reference ctor (same RNG consumption, same key order)
        self.A = nn.Parameter(A.clone())
        self.pre = nn.Sequential(nn.Conv2d(in_channels, mid * K, 1), _norm_layer(norm, mid * K), nn.ReLU())
        self.post = nn.Conv2d(mid * K, out_channels, 1)
        self.alpha = nn.Parameter(torch.zeros(K))
        self.beta = nn.Parameter(torch.zeros(K))
        self.conv1 = nn.Conv2d(in_channels, mid * K, 1)
        self.conv2 = nn.Conv2d(in_channels, mid * K, 1)
        if in_channels != out_channels:
            self.down = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1), _norm_layer(norm, out_channels))
        else:
            self.down = None
        self.bn = _norm_layer(norm, out_channels)
        V = A.size(-1)
        # constants that turn K-B's typed slots into plain ones: one node type, one edge class, identity edge linear
        self.register_buffer('_nt0', torch.zeros(V, dtype=torch.int32), persistent=False)
        self.register_buffer('_et0', torch.zeros(V, V, dtype=torch.int32), persistent=False)
        self.register_buffer('_we_eye', torch.eye(mid), persistent=False)

    def fusable_pairs(self):
        return [(self.post, self.bn)]

    def adjacency(self, xbar):
        """Ahat (n, K*mid, V, V) from the time-averaged input xbar (n, Ci, V)."""
        m, Ci = self.mid_channels, self.in_channels
        w1, b1 = self.conv1.weight.flatten(1), self.conv1.bias
        w2, b2 = self.conv2.weight.flatten(1), self.conv2.bias
        zw, zb = w1.new_zeros(m, Ci), b1.new_zeros(m)
        we, be = self._we_eye, zb
        a = self.alpha if self.subset_wise else self.alpha[0].expand(3)
        b = self.beta if self.subset_wise else self.beta[0].expand(3)
        dyn = kernels.ops().dynadj
        # subsets 0 and 1: slots (a0, b0) and (a1, b1) of one call (its third, node-typed slot is fed zeros and dropped)
        first = dyn(xbar, self.A, a, b, w1[:2 * m], b1[:2 * m], w2[:2 * m], b2[:2 * m], zw, zb, we, be, self._nt0, self._et0,
                    single_use=False)
        # subset 2: slot (a0, b0) of a second call
        A2 = torch.cat([self.A[2:3], torch.zeros_like(self.A[:2])])
        a2 = torch.cat([a[2:3], a.new_zeros(2)])
        b2s = torch.cat([b[2:3], b.new_zeros(2)])
        second = dyn(xbar, A2, a2, b2s, torch.cat([w1[2 * m:], zw]), torch.cat([b1[2 * m:], zb]),
                     torch.cat([w2[2 * m:], zw]), torch.cat([b2[2 * m:], zb]), zw, zb, we, be, self._nt0, self._et0,
                     single_use=False)
        return torch.cat([first[:, :2 * m], second[:, :m]], 1)

    def forward_deferred(self, x, xbar=None, x_res=None):
        ops = kernels.ops()
        x_res = x if x_res is None else x_res
        if xbar is None:
            xbar = ops.tmean(x)
        ahat = self.adjacency(xbar)
        zp, _, ap = conv_bn(x, None, None, None, False, self.pre[0], 1, False, self.pre[1])
        y = ops.aggregate(zp, ap, True, ahat)
        zo, _, ao = conv_bn(y, None, None, None, False, self.post, 1, False, self.bn)
        if self.down is None:
            return Deferred(zo, ao, x_res, None, True)
        zd, _, ad = conv_bn(x_res, None, None, None, False, self.down[0], 1, False, self.down[1])
        return Deferred(zo, ao, zd, ad, True)

    def forward(self, x, A=None):
        out = self.forward_deferred(x).materialize()
        flush_running_stats()
        return out


class unit_aagcn(nn.Module):
    """2s-AGCN / AAGCN spatial unit (reference: pyskl/models/gcns/utils/gcn.py:349-460): per subset a data-dependent
    topology ``A_i + alpha * tanh(conv_a_i(x)^T conv_b_i(x) / (inter_c * T))`` per sample, ``y = sum_i conv_d_i(x A_i)``,
    BN, + down(x), ReLU, then the spatial / temporal / channel attention gates (``y * sigmoid(.) + y`` each).
    HIP chain: the embedding convs as two K-C launches (all conv_a, all conv_b), their Gram on K-A''s backward product,
    the three ``conv_d`` as ONE K-C launch BEFORE the aggregation (the 1x1 conv and the joint mixing commute), the
    aggregation with the per-sample topologies summed over subsets inside K-A' (BN statistics in its epilogue), ``down`` +
    BN + ReLU fused into the output pass, the three gates as one pass each (csrc/aagcn.hip) with the next gate's mean in the
    same launch.  Left to PyTorch: tanh / sigmoid and the convs / linears on KB-sized (N, C, V)-class tensors."""

    def __init__(self, in_channels, out_channels, A, coff_embedding=4, adaptive=True, attention=True):
        super().__init__()
        inter_channels = out_channels // coff_embedding
        self.inter_c = inter_channels
        self.out_c = out_channels
        self.in_c = in_channels
        self.num_subset = A.shape[0]
        self.adaptive = adaptive
        self.attention = attention
        num_joints = A.shape[-1]
        # parameter creation order follows the reference ctor (same RNG consumption, same key order)
        self.conv_d = nn.ModuleList([nn.Conv2d(in_channels, out_channels, 1) for _ in range(self.num_subset)])
        if self.adaptive:
            self.A = nn.Parameter(A)
            self.alpha = nn.Parameter(torch.zeros(1))
            self.conv_a = nn.ModuleList()
            self.conv_b = nn.ModuleList()
            for _ in range(self.num_subset):
                self.conv_a.append(nn.Conv2d(in_channels, inter_channels, 1))
                self.conv_b.append(nn.Conv2d(in_channels, inter_channels, 1))
        else:
            self.register_buffer('A', A)
        if self.attention:
            self.conv_ta = nn.Conv1d(out_channels, 1, 9, padding=4)
            ker_joint = num_joints if num_joints % 2 else num_joints - 1
            self.conv_sa = nn.Conv1d(out_channels, 1, ker_joint, padding=(ker_joint - 1) // 2)
            self.fc1c = nn.Linear(out_channels, out_channels // 2)
            self.fc2c = nn.Linear(out_channels // 2, out_channels)
        if in_channels != out_channels:
            self.down = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1), nn.BatchNorm2d(out_channels))
        else:
            self.down = None
        self.bn = nn.BatchNorm2d(out_channels)

    def init_weights(self):
        # reference: gcn.py:401-423 (same order of random draws)
        import math
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out')
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.bn.weight, 1e-6)
        nn.init.constant_(self.bn.bias, 0)
        for conv in self.conv_d:
            w = conv.weight
            nn.init.normal_(w, 0, math.sqrt(2. / (w.size(0) * w.size(1) * w.size(2) * self.num_subset)))
            nn.init.constant_(conv.bias, 0)
        if self.attention:
            nn.init.constant_(self.conv_ta.weight, 0)
            nn.init.constant_(self.conv_ta.bias, 0)
            nn.init.xavier_normal_(self.conv_sa.weight)
            nn.init.constant_(self.conv_sa.bias, 0)
            nn.init.kaiming_normal_(self.fc1c.weight)
            nn.init.constant_(self.fc1c.bias, 0)
            nn.init.constant_(self.fc2c.weight, 0)
            nn.init.constant_(self.fc2c.bias, 0)

    def forward_deferred(self, x, xbar=None, x_res=None):
        ops = kernels.ops()
        x_res = x if x_res is None else x_res
        N, C, T, V = x.shape
        S, ic, Co = self.num_subset, self.inter_c, self.out_c
        if self.adaptive:
            # embeddings: two K-C launches (all conv_a, all conv_b) so that each side is one contiguous (N*S, ic, T, V)
            wa = torch.cat([m.weight.flatten(1) for m in self.conv_a], 0)
            wb = torch.cat([m.weight.flatten(1) for m in self.conv_b], 0)
            ba = torch.cat([m.bias for m in self.conv_a], 0)
            bb = torch.cat([m.bias for m in self.conv_b], 0)
            ea = ops.pwconv(x, None, None, None, False, wa, ba, 1, False)[0].reshape(N * S, ic, T, V)
            eb = ops.pwconv(x, None, None, None, False, wb, bb, 1, False)[0].reshape(N * S, ic, T, V)
            gram = ops.gram(ea, eb).view(N, S, V, V) / (ic * T)                              # gcn.py:432-434
            adj = self.A[None] + torch.tanh(gram) * self.alpha                               # gcn.py:435 (KB-sized)
        else:
            adj = self.A
        # sum_i conv_d_i(x . A_i) = sum_i (W_i x) . A_i: the 1x1 conv acts on channels, A_i on joints, so the conv runs
        # first, as ONE launch over the stacked weights, and the sum over subsets happens inside K-A''s accumulators
        # (with the BN statistics in its epilogue).  The biases commute only as their sum: it is a per-channel constant
        # under the BatchNorm that follows — it moves the running mean, not the output.
        wd = torch.cat([m.weight.flatten(1) for m in self.conv_d], 0)                        # (S*Co, Ci)
        bsum = torch.stack([m.bias for m in self.conv_d]).sum(0)
        p = ops.pwconv(x, None, None, None, False, wd, None, 1, False)[0]                    # (N, S*Co, T, V)
        if _need_stats(self.bn):
            zo, sc, sh, mean, var = ops.aggregate_sum(p, adj, S, self.bn.weight, self.bn.bias, self.bn.eps, True,
                                                      self.adaptive)
            record_running(self.bn, mean + bsum.detach(), var, N * T * V)
            # (keeps the biases on the graph: their gradient is exactly zero here.  The scale goes on as a fresh alias: the
            # BatchNorm context that rides on `sc` would let the consumer hand its sums straight to the producer and return
            # no gradient for the shift — the biases would end up without one)
            ao = (sc.view_as(sc), sh + 0.0 * bsum)
        else:
            zo = ops.aggregate_sum(p, adj, S, per_sample=self.adaptive)[0]
            sc, sh = eval_affine(self.bn)
            ao = (sc, sh + sc * bsum)
        want_mean = bool(self.attention)
        if self.down is None:
            y, r0 = ops.fuse_out(zo, ao, x_res, None, True, want_mean)
        else:
            zd, _, ad = conv_bn(x_res, None, None, None, False, self.down[0], 1, False, self.down[1])
            y, r0 = ops.fuse_out(zo, ao, zd, ad, True, want_mean)
        if self.attention:                                                                   # gcn.py:447-459
            # three gates y <- y * sigmoid(.) + y; each pass also emits the mean the next gate is computed from, the
            # KB-sized convs / linears on those means stay PyTorch (like the head)
            g1 = torch.sigmoid(self.conv_sa(r0)).squeeze(1)                                  # (N, V) from the mean over T
            y, r1 = ops.gate(y, g1, 0, 1)
            g2 = torch.sigmoid(self.conv_ta(r1)).squeeze(1)                                  # (N, T) from the mean over V
            y, r2 = ops.gate(y, g2, 1, 2)
            g3 = torch.sigmoid(self.fc2c(torch.relu(self.fc1c(r2))))                         # (N, C)
            y, _ = ops.gate(y, g3, 2, 0)
        return as_deferred(y)

    def forward(self, x):
        out = self.forward_deferred(x).materialize()
        flush_running_stats()
        return out


def _pw_bn(ops, z, w, b, gamma, beta, eps, want):
    """pwconv with the BN statistics of its output -> (out, scale, shift, mean, var) (the op_bn contract)"""
    if want:
        out, _, sc, sh, mean, var = ops.pwconv(z, None, None, None, False, w, b, 1, False, gamma, beta, eps, w.shape[0], True)
        return out, sc, sh, mean, var
    return (ops.pwconv(z, None, None, None, False, w, b, 1, False)[0], None, None, None, None)


class unit_gcn(nn.Module):
    """ST-GCN spatial unit (reference: gcn.py:22-97).  ``conv_pos='pre'``: conv Ci->K*Co, aggregate with the K learnable /
    fixed adjacencies summed over subsets (K-A'), BN, (+res), ReLU.  ``conv_pos='post'``: aggregate the input with each
    adjacency first (K launches of K-A' on one subset each), then one conv over the K*Ci stacked channels.
    ``adaptive``: None / 'init' use ``A`` (buffer / parameter); 'offset' uses ``A + PA``, 'importance' ``A * PA`` with the
    extra parameter ``PA`` (gcn.py:54-60,83)."""

    def __init__(self, in_channels, out_channels, A, adaptive='init', conv_pos='pre', with_res=False, norm='BN',
                 act='ReLU'):
        super().__init__()
        assert adaptive in [None, 'init', 'offset', 'importance']
        assert conv_pos in ['pre', 'post']
        _check_act(act)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.num_subsets = A.size(0)
        self.adaptive = adaptive
        self.conv_pos = conv_pos
        self.with_res = with_res
        self.bn = _norm_layer(norm, out_channels)
        if adaptive == 'init':
            self.A = nn.Parameter(A.clone())
        else:
            self.register_buffer('A', A)
        if adaptive in ('offset', 'importance'):
            self.PA = nn.Parameter(A.clone())
            if adaptive == 'offset':
                nn.init.uniform_(self.PA, -1e-6, 1e-6)
            else:
                nn.init.constant_(self.PA, 1)
        if conv_pos == 'pre':
            self.conv = nn.Conv2d(in_channels, out_channels * A.size(0), 1)
        else:
            self.conv = nn.Conv2d(A.size(0) * in_channels, out_channels, 1)
        self.down = None
        if with_res and in_channels != out_channels:
            self.down = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1), _norm_layer(norm, out_channels))

    def fusable_pairs(self):
        # conv_pos='pre': the aggregate sits between conv and bn (the bias does not commute with it): nothing to fold
        return [(self.conv, self.bn)] if self.conv_pos == 'post' else []

    def effective_A(self):
        if self.adaptive == 'offset':
            return self.A + self.PA
        if self.adaptive == 'importance':
            return self.A * self.PA
        return self.A

    def forward_deferred(self, x, x_res=None):
        ops = kernels.ops()
        x_res = x if x_res is None else x_res
        A = self.effective_A()                                   # (K, V, V): a KB-sized elementwise op on parameters
        K = self.num_subsets
        if self.conv_pos == 'pre':
            h = ops.pwconv(x, None, None, None, False, self.conv.weight, self.conv.bias, 1, False)[0]
            y, ay = op_bn(self.bn, lambda g, b, eps, want: ops.aggregate_sum(h, A, K, g, b, eps, want),
                          lambda y: y.shape[0] * y.shape[2] * y.shape[3])
        else:
            parts = [ops.aggregate_sum(x, A[k:k + 1], 1)[0] for k in range(K)]        # x . A_k, one subset per launch
            y, _, ay = conv_bn(torch.cat(parts, 1), None, None, None, False, self.conv, 1, False, self.bn)
        if not self.with_res:
            return Deferred(y, ay, None, None, True)
        if self.down is None:
            return Deferred(y, ay, x_res, None, True)
        zd, _, ad = conv_bn(x_res, None, None, None, False, self.down[0], 1, False, self.down[1])
        return Deferred(y, ay, zd, ad, True)

    def forward(self, x, A=None):
        if A is not None:
            raise NotImplementedError('passing A at call time (reference quirk Q9) is not supported')
        out = self.forward_deferred(x).materialize()
        flush_running_stats()
        return out

    def init_weights(self):
        pass


def _kaiming_conv_init(module):
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out')
            nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


class CTRGC(nn.Module):
    """Channel-wise topology refinement conv of one subset (reference: gcn.py:634-666).  Owns the parameters; the
    arithmetic of the K subsets runs batched in ``unit_ctrgcn.forward_deferred``."""

    def __init__(self, in_channels, out_channels, rel_reduction=8):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.rel_channels = 8 if in_channels <= 16 else in_channels // rel_reduction
        self.conv1 = nn.Conv2d(in_channels, self.rel_channels, kernel_size=1)
        self.conv2 = nn.Conv2d(in_channels, self.rel_channels, kernel_size=1)
        self.conv3 = nn.Conv2d(in_channels, out_channels, kernel_size=1)
        self.conv4 = nn.Conv2d(self.rel_channels, out_channels, kernel_size=1)
        self.tanh = nn.Tanh()
        self.init_weights()

    def init_weights(self):
        _kaiming_conv_init(self)


class unit_ctrgcn(nn.Module):
    """CTR-GCN spatial unit (reference: gcn.py:882-929): per subset k a refined topology
    ``Ahat_k = alpha*conv4_k(tanh(conv1_k(x).mean(T)[u] - conv2_k(x).mean(T)[v])) + A[k]`` per sample and output
    channel, ``y = sum_k aggregate(conv3_k(x), Ahat_k)``, BN, + down(x), ReLU.  HIP chain: the six mean-pooled
    projections are one K-C launch on the time mean (the mean commutes with the 1x1 conv), conv3 of the three subsets
    is one K-C launch, the sum over subsets happens inside the aggregate's accumulators (K-A')."""

    def __init__(self, in_channels, out_channels, A):
        super().__init__()
        self.inter_c = out_channels // 4
        self.out_c = out_channels
        self.in_c = in_channels
        self.num_subset = A.shape[0]
        self.convs = nn.ModuleList([CTRGC(in_channels, out_channels) for _ in range(self.num_subset)])
        if in_channels != out_channels:
            self.down = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1), nn.BatchNorm2d(out_channels))
        else:
            self.down = None
        self.A = nn.Parameter(A.clone())
        self.alpha = nn.Parameter(torch.zeros(1))
        self.bn = nn.BatchNorm2d(out_channels)
        self.soft = nn.Softmax(-2)
        self.relu = nn.ReLU(inplace=True)

    def flat_groups(self):
        """(see dgmstcn.flat_groups) the per-subset projections run as stacked convs: conv1 and conv2 of all subsets back to
        back (ctr_topology concatenates the two stacks again), conv3 of all subsets"""
        cs = self.convs
        return [[c.conv1.weight for c in cs] + [c.conv2.weight for c in cs], [c.conv1.bias for c in cs] + [c.conv2.bias for c in cs],
                [c.conv3.weight for c in cs], [c.conv3.bias for c in cs]]

    def forward_deferred(self, x, xbar=None, x_res=None):
        ops = kernels.ops()
        x_res = x if x_res is None else x_res
        if xbar is None:
            xbar = ops.tmean(x)
        cs = self.convs
        cat = kernels.cat_rows                      # views of the flat parameter buffer under FlatParams (flat_groups)
        ahat = ops.ctr_topology(
            xbar, cat([c.conv1.weight.flatten(1) for c in cs]), cat([c.conv1.bias for c in cs]),
            cat([c.conv2.weight.flatten(1) for c in cs]), cat([c.conv2.bias for c in cs]),
            [c.conv4.weight.flatten(1) for c in cs], [c.conv4.bias for c in cs], self.alpha, self.A, subset_major=True)
        w3 = cat([c.conv3.weight.flatten(1) for c in cs])
        b3 = cat([c.conv3.bias for c in cs])
        x3 = ops.pwconv(x, None, None, None, False, w3, b3, 1, False)[0]
        y, ay = op_bn(self.bn, lambda g, b, eps, want: ops.aggregate_sum(x3, ahat, self.num_subset, g, b, eps, want),
                      lambda y: y.shape[0] * y.shape[2] * y.shape[3])
        if self.down is None:
            return Deferred(y, ay, x_res, None, True)
        zd, _, ad = conv_bn(x_res, None, None, None, False, self.down[0], 1, False, self.down[1])
        return Deferred(y, ay, zd, ad, True)

    def forward(self, x):
        out = self.forward_deferred(x).materialize()
        flush_running_stats()
        return out

    def init_weights(self):
        _kaiming_conv_init(self)
        nn.init.constant_(self.bn.weight, 1e-6)
        nn.init.constant_(self.bn.bias, 0)


class CTRHGC(nn.Module):
    """One subset of the heterogeneous CTR unit (reference: gcn.py:668-771).  Owns the parameters; the arithmetic of the K
    subsets runs batched in ``unit_ctrhgcn.forward_deferred``.  Supported flag set = what the shipped CTR-GCN config
    reaches (configs/ctrgcn/CTRGCN_model.py through unit_ctrhgcn's per-subset overrides): no node attention, optional
    edge attention (reduced channels), optional ``ada`` Gram term; target_specific / full_channels / add_type raise."""

    def __init__(self, in_channels, out_channels, rel_reduction=8, node_attention=True, edge_attention=False,
                 target_specific=False, full_channels=False, add_type=False, ada=False, num_types=5, edge_num=15,
                 semantic_index=False):
        super().__init__()
        if (node_attention and semantic_index) or target_specific or full_channels or add_type:
            raise NotImplementedError('CTRHGC: node attention / target_specific / full_channels / add_type have no HIP path')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.node_attention, self.edge_attention = node_attention, edge_attention
        self.num_types, self.edge_num, self.ada, self.semantic_index = num_types, edge_num, ada, semantic_index
        if ada:
            self.beta = nn.Parameter(torch.zeros(1))
        self.rel_channels = 8 if in_channels <= 16 else in_channels // rel_reduction
        self.conv1 = nn.Conv2d(in_channels, self.rel_channels, kernel_size=1)
        self.conv2 = nn.Conv2d(in_channels, self.rel_channels, kernel_size=1)
        if edge_attention and semantic_index:
            self.edge_att_conv = nn.Conv2d(self.rel_channels, edge_num * self.rel_channels, 1)
        self.conv4 = nn.Conv2d(self.rel_channels, out_channels, kernel_size=1)
        self.conv3 = nn.Conv2d(in_channels, out_channels, 1)
        self.tanh = nn.Tanh()
        _kaiming_conv_init(self)


class unit_ctrhgcn(nn.Module):
    """Heterogeneous CTR-GCN spatial unit (reference: gcn.py:773-880): like ``unit_ctrgcn`` with a learnable scale per
    subset, an edge-typed attention conv on the first subset (each joint pair keeps the variant of its edge class) and
    the Gram term ``beta_k * x1_k^T x2_k`` (``ada``).  Quirk kept: the constructor's per-subset overrides leave node
    attention off everywhere and edge attention on for subset 0 only (gcn.py:801-842)."""

    def __init__(self, in_channels, out_channels, A, edge_type, node_type, semantic_index=False, rel_reduction=8,
                 node_attention=False, edge_attention=False, target_specific=False, full_channels=False, add_type=False,
                 ada=False, num_types=5, edge_num=15):
        super().__init__()
        self.inter_c, self.out_c, self.in_c = out_channels // 4, out_channels, in_channels
        self.num_subset = A.shape[0]
        self.register_buffer('edge_type_idx', torch.as_tensor(edge_type).to(torch.int32).reshape(-1).contiguous(),
                             persistent=False)
        self.convs = nn.ModuleList()
        for i in range(self.num_subset):
            if i == 0:
                node_attention = False
            if i >= 1:
                edge_attention = False
            if i == 2:
                node_attention = False
            if i <= 2:
                self.convs.append(CTRHGC(in_channels, out_channels, rel_reduction=rel_reduction,
                                         node_attention=node_attention, edge_attention=edge_attention,
                                         target_specific=target_specific, full_channels=full_channels, add_type=add_type,
                                         ada=ada, num_types=num_types, edge_num=edge_num, semantic_index=semantic_index))
        if in_channels != out_channels:
            self.down = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1), nn.BatchNorm2d(out_channels))
        else:
            self.down = None
        self.A = nn.Parameter(A.clone())
        self.alpha = nn.Parameter(torch.zeros(self.A.size(0)))
        self.bn = nn.BatchNorm2d(out_channels)
        self.soft = nn.Softmax(-2)
        self.relu = nn.ReLU(inplace=True)

    def flat_groups(self):
        """(see dgmstcn.flat_groups) the per-subset projections run as stacked convs: conv1 and conv2 of all subsets back to
        back (ctr_topology concatenates the two stacks again), conv3 of all subsets"""
        cs = self.convs
        return [[c.conv1.weight for c in cs] + [c.conv2.weight for c in cs], [c.conv1.bias for c in cs] + [c.conv2.bias for c in cs],
                [c.conv3.weight for c in cs], [c.conv3.bias for c in cs]]

    def forward_deferred(self, x, xbar=None, x_res=None):
        ops = kernels.ops()
        x_res = x if x_res is None else x_res
        if xbar is None:
            xbar = ops.tmean(x)
        cs = self.convs
        edge = {k: (c.edge_att_conv.weight.flatten(1), c.edge_att_conv.bias, self.edge_type_idx)
                for k, c in enumerate(cs) if hasattr(c, 'edge_att_conv')}
        beta = torch.cat([c.beta for c in cs]) if all(c.ada for c in cs) else None
        cat = kernels.cat_rows
        ahat = ops.ctr_topology(
            xbar, cat([c.conv1.weight.flatten(1) for c in cs]), cat([c.conv1.bias for c in cs]),
            cat([c.conv2.weight.flatten(1) for c in cs]), cat([c.conv2.bias for c in cs]),
            [c.conv4.weight.flatten(1) for c in cs], [c.conv4.bias for c in cs], self.alpha, self.A, beta, edge)
        w3 = cat([c.conv3.weight.flatten(1) for c in cs])
        b3 = cat([c.conv3.bias for c in cs])
        x3 = ops.pwconv(x, None, None, None, False, w3, b3, 1, False)[0]
        y, ay = op_bn(self.bn, lambda g, b, eps, want: ops.aggregate_sum(x3, ahat, self.num_subset, g, b, eps, want),
                      lambda y: y.shape[0] * y.shape[2] * y.shape[3])
        if self.down is None:
            return Deferred(y, ay, x_res, None, True)
        zd, _, ad = conv_bn(x_res, None, None, None, False, self.down[0], 1, False, self.down[1])
        return Deferred(y, ay, zd, ad, True)

    def forward(self, x):
        out = self.forward_deferred(x).materialize()
        flush_running_stats()
        return out

    def init_weights(self):
        _kaiming_conv_init(self)
        nn.init.constant_(self.bn.weight, 1e-6)
        nn.init.constant_(self.bn.bias, 0)
