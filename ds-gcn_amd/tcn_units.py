"""Temporal units behind the reference's class names and ctor kwargs.

``dgmstcn`` (reference: pyskl/models/gcns/utils/tcn.py:344-431): the six branch 1x1 convs share
their input, so they run as ONE C->C channel mix (weights concatenated) whose epilogue also emits
the global-joint column (mean over V — by linearity of the 1x1 conv, tcn.py:409) and the batch
statistics; one K-D kernel then does BN+ReLU, the four dilated 3-tap convs, the max-pool, the
strided pass-through and the local + global*add_coeff combine (tcn.py:416-420).
``unit_tcn`` (tcn.py:10-37): k=1 (block residual) is a strided channel mix; k>1 the dense temporal conv.
"""
import contextlib

import torch
import torch.nn as nn

from . import kernels
from .gcn_units import (Deferred, as_deferred, conv_bn, eval_affine, flush_running_stats, op_bn, record_running,
                        _need_stats, _norm_layer)


class unit_tcn(nn.Module):

    def __init__(self, in_channels, out_channels, kernel_size=9, stride=1, dilation=1, norm='BN', dropout=0):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.dilation = dilation
        pad = (kernel_size + (kernel_size - 1) * (dilation - 1) - 1) // 2
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=(kernel_size, 1), padding=(pad, 0),
                              stride=(stride, 1), dilation=(dilation, 1))
        self.bn = _norm_layer(norm, out_channels) if norm is not None else nn.Identity()
        self.drop = nn.Dropout(dropout, inplace=True)
        self.stride = stride

    def fusable_pairs(self):
        return [(self.conv, self.bn)]

    def _gemm_ok(self, d):
        """Does the GEMM form take this shape?  (asked before op_bn records running statistics)"""
        n, Ci, T, V = d.x1.shape
        fn = getattr(kernels.ops(), 'tconv_gemm_ok', None)
        return True if fn is None else fn(n, Ci, self.out_channels, T, V, self.kernel_size, self.stride)

    def forward_deferred(self, x):
        """x: tensor or Deferred -> Deferred(z, bn affine) (dropout must be 0 on this path)."""
        ops = kernels.ops()
        pre = x if isinstance(x, kernels.Prestrided) else None
        if pre is not None:
            if self.kernel_size != 1 or pre.stride != self.stride:
                raise ValueError(f'unit_tcn(kernel_size={self.kernel_size}, stride={self.stride}) was handed frames '
                                 f'pre-strided by {pre.stride}')
            x = pre.x
        d = as_deferred(x)
        has_bn = isinstance(self.bn, nn.BatchNorm2d)
        if self.kernel_size == 1:
            stride = self.stride
            if pre is not None:
                stride = 1                      # the producer already handed over the kept frames (kernels._FuseOut, tee = 2)
            elif stride > 1 and d.a1 is None and d.x2 is None and not d.relu and hasattr(ops, 'strided_frames'):
                # a plain input (the block residual): pick the kept frames first (one strided-copy launch, tapconv's
                # pass-through window) and run the channel mix on the compact tensor — the strided 1x1 conv has only the
                # scalar-load kernels (planes of every other 25-joint row are not 16-byte runs): 290 -> ~125 us per block
                d = Deferred(ops.strided_frames(d.x1, stride), None, None, None, False)
                stride = 1
            if has_bn:
                z, _, az = conv_bn(d.x1, d.a1, d.x2, d.a2, d.relu, self.conv, stride, False, self.bn)
                return Deferred(z, az, None, None, False)
            z = ops.pwconv(d.x1, d.a1, d.x2, d.a2, d.relu, self.conv.weight, self.conv.bias, stride, False)[0]
            return Deferred(z, None, None, None, False)
        w, b = self.conv.weight, self.conv.bias
        if self.stride in (1, 2) and self.dilation == 1 and hasattr(ops, 'tconv_bn') and self._gemm_ok(d):
            # GEMM form (csrc/tcg.hip): the virtual input is activated while loading, statistics in the epilogue
            if not has_bn:
                return Deferred(ops.tconv_bn(d.x1, d.a1, d.x2, d.a2, d.relu, w, b, stride=self.stride)[0], None, None, None,
                                False)
            z, az = op_bn(self.bn, lambda g, be, eps, want: ops.tconv_bn(d.x1, d.a1, d.x2, d.a2, d.relu, w, b, g, be, eps,
                                                                          want, self.stride),
                          lambda z: z.shape[0] * z.shape[2] * z.shape[3])
            return Deferred(z, az, None, None, False)
        # otherwise the dense temporal conv reads a materialised tensor (zero padding applies to the activated values)
        h = d.x1 if (d.a1 is None and d.x2 is None and not d.relu) else d.materialize()
        if not has_bn:
            return Deferred(ops.tconv(h, w, b, self.stride, self.dilation)[0], None, None, None, False)
        z, az = op_bn(self.bn, lambda g, be, eps, want: ops.tconv(h, w, b, self.stride, self.dilation, g, be, eps, want),
                      lambda z: z.shape[0] * z.shape[2] * z.shape[3])
        return Deferred(z, az, None, None, False)

    def forward(self, x):
        out = self.forward_deferred(x).materialize()
        flush_running_stats()
        return self.drop(out) if self.drop.p > 0 else out

    def init_weights(self):
        nn.init.kaiming_normal_(self.conv.weight, mode='fan_out')
        nn.init.constant_(self.conv.bias, 0)
        if isinstance(self.bn, nn.BatchNorm2d):
            nn.init.constant_(self.bn.weight, 1)
            nn.init.constant_(self.bn.bias, 0)


class dgmstcn(nn.Module):

    def __init__(self, in_channels, out_channels, mid_channels=None, num_joints=25, dropout=0.,
                 ms_cfg=[(3, 1), (3, 2), (3, 3), (3, 4), ('max', 3), '1x1'], stride=1):
        super().__init__()
        self.ms_cfg = [tuple(c) if isinstance(c, (list, tuple)) else c for c in ms_cfg]
        num_branches = len(ms_cfg)
        self.num_branches = num_branches
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.stride = stride
        self.act = nn.ReLU()
        self.num_joints = num_joints
        self.add_coeff = nn.Parameter(torch.zeros(self.num_joints))
        if mid_channels is None:
            mid_channels = out_channels // num_branches
            rem_mid_channels = out_channels - mid_channels * (num_branches - 1)
        else:
            assert isinstance(mid_channels, float) and mid_channels > 0
            mid_channels = int(out_channels * mid_channels)
            rem_mid_channels = mid_channels
        self.mid_channels = mid_channels
        self.rem_mid_channels = rem_mid_channels

        branches = []
        self.widths = []
        for i, cfg in enumerate(self.ms_cfg):
            bc = rem_mid_channels if i == 0 else mid_channels
            self.widths.append(bc)
            if cfg == '1x1':
                branches.append(nn.Conv2d(in_channels, bc, kernel_size=1, stride=(stride, 1)))
                continue
            assert isinstance(cfg, tuple)
            if cfg[0] == 'max':
                branches.append(nn.Sequential(
                    nn.Conv2d(in_channels, bc, kernel_size=1), nn.BatchNorm2d(bc), self.act,
                    nn.MaxPool2d(kernel_size=(cfg[1], 1), stride=(stride, 1), padding=(1, 0))))
                continue
            assert isinstance(cfg[0], int) and isinstance(cfg[1], int)
            branches.append(nn.Sequential(
                nn.Conv2d(in_channels, bc, kernel_size=1), nn.BatchNorm2d(bc), self.act,
                unit_tcn(bc, bc, kernel_size=cfg[0], stride=stride, dilation=cfg[1], norm=None)))
        self.branches = nn.ModuleList(branches)
        tin_channels = mid_channels * (num_branches - 1) + rem_mid_channels
        self.transform = nn.Sequential(nn.BatchNorm2d(tin_channels), self.act,
                                       nn.Conv2d(tin_channels, out_channels, kernel_size=1))
        self.bn = nn.BatchNorm2d(out_channels)
        self.drop = nn.Dropout(dropout, inplace=True)
        # HIP K-D layout: BN+ReLU branches first, pass-through ('1x1') branches last
        seen_plain = False
        for cfg in self.ms_cfg:
            if cfg == '1x1':
                seen_plain = True
            elif seen_plain:
                raise NotImplementedError("dgmstcn: '1x1' branches must come last in ms_cfg on the HIP path")
        self.n_act = sum(w for w, c in zip(self.widths, self.ms_cfg) if c != '1x1')

    def _first_convs(self):
        return [b if isinstance(b, nn.Conv2d) else b[0] for b in self.branches]

    def fusable_pairs(self):
        return [(self.transform[2], self.bn)]

    def flat_groups(self):
        """Parameter tensors the forward concatenates: FlatParams lays each group out back to back, so the
        concatenation is a view (kernels.cat_rows)."""
        convs = self._first_convs()
        bns = [b[1] for b in self.branches if not isinstance(b, nn.Conv2d)]
        return [[c.weight for c in convs], [c.bias for c in convs], [bn.weight for bn in bns], [bn.bias for bn in bns]]

    def forward_deferred(self, g, hold=None):
        """g: tensor or Deferred (the gcn output) -> Deferred(zt, affine of self.bn).  hold: a kernels.bn_batch that already
        holds the finalize job of an independent conv (the block's residual conv, issued first): the transform conv's job
        joins it and the two go out as one launch."""
        ops = kernels.ops()
        d = as_deferred(g)
        n, _, T, V = d.x1.shape
        convs = self._first_convs()
        wb = kernels.cat_rows([c.weight.flatten(1) for c in convs])
        bb = kernels.cat_rows([c.bias for c in convs])
        bns = [b[1] for b in self.branches if not isinstance(b, nn.Conv2d)]
        train_stats = any(_need_stats(bn) for bn in bns)
        count = n * T * (V + 1)
        if train_stats:
            gamma = kernels.cat_rows([bn.weight for bn in bns])
            beta = kernels.cat_rows([bn.bias for bn in bns])
            z, zaug, scale, shift, m, var = ops.pwconv(d.x1, d.a1, d.x2, d.a2, d.relu, wb, bb, 1, True, gamma, beta,
                                                      bns[0].eps, self.n_act, True)
            c0 = 0
            for bn in bns:
                record_running(bn, m[c0:c0 + bn.num_features], var[c0:c0 + bn.num_features], count)
                c0 += bn.num_features
        else:
            z, zaug = ops.pwconv(d.x1, d.a1, d.x2, d.a2, d.relu, wb, bb, 1, True)[:2]
            aff = [eval_affine(bn) for bn in bns]
            rest = self.out_channels - self.n_act
            scale = torch.cat([a[0] for a in aff] + ([z.new_ones(rest)] if rest else []))
            shift = torch.cat([a[1] for a in aff] + ([z.new_zeros(rest)] if rest else []))
        tconvs = [b[3].conv for b in self.branches if not isinstance(b, nn.Conv2d) and isinstance(b[3], unit_tcn)]
        bn1 = self.transform[0]
        tw, tb = [c.weight for c in tconvs], [c.bias for c in tconvs]
        if _need_stats(bn1):
            f, s1, h1, m1, v1 = ops.temporal_ms(z, zaug, scale, shift, self.n_act, self.ms_cfg, self.widths, tw, tb,
                                                self.add_coeff, self.stride, bn1.weight, bn1.bias, bn1.eps, True)
            record_running(bn1, m1, v1, f.shape[0] * f.shape[2] * f.shape[3])
            a1 = (s1, h1)
        else:
            f = ops.temporal_ms(z, zaug, scale, shift, self.n_act, self.ms_cfg, self.widths, tw, tb, self.add_coeff,
                                self.stride)[0]
            a1 = eval_affine(bn1)
        with (hold if hold is not None else contextlib.nullcontext()):
            zt, _, a2 = conv_bn(f, a1, None, None, True, self.transform[2], 1, False, self.bn)
        return Deferred(zt, a2, None, None, False)

    def forward(self, x):
        out = self.forward_deferred(x).materialize()
        flush_running_stats()
        return self.drop(out) if self.drop.p > 0 else out

    def init_weights(self):
        pass


class mstcn(dgmstcn):
    """ST-GCN++'s multi-scale temporal unit (reference: tcn.py:104-177) = ``dgmstcn`` without the global joint: same
    branches, ``cat`` -> BN -> ReLU -> 1x1 -> BN.  Same constructor (minus ``num_joints``) and state_dict keys."""

    def __init__(self, in_channels, out_channels, mid_channels=None, dropout=0.,
                 ms_cfg=[(3, 1), (3, 2), (3, 3), (3, 4), ('max', 3), '1x1'], stride=1):
        super().__init__(in_channels, out_channels, mid_channels=mid_channels, dropout=dropout, ms_cfg=ms_cfg,
                         stride=stride)
        del self.add_coeff            # not a parameter of the reference's mstcn
        del self.num_joints

    def forward_deferred(self, g):
        ops = kernels.ops()
        d = as_deferred(g)
        n, _, T, V = d.x1.shape
        convs = self._first_convs()
        wb = kernels.cat_rows([c.weight.flatten(1) for c in convs])
        bb = kernels.cat_rows([c.bias for c in convs])
        bns = [b[1] for b in self.branches if not isinstance(b, nn.Conv2d)]
        if any(_need_stats(bn) for bn in bns):
            gamma = kernels.cat_rows([bn.weight for bn in bns])
            beta = kernels.cat_rows([bn.bias for bn in bns])
            z, _, scale, shift, m, var = ops.pwconv(d.x1, d.a1, d.x2, d.a2, d.relu, wb, bb, 1, False, gamma, beta,
                                                   bns[0].eps, self.n_act, True)
            c0 = 0
            for bn in bns:
                record_running(bn, m[c0:c0 + bn.num_features], var[c0:c0 + bn.num_features], n * T * V)
                c0 += bn.num_features
        else:
            z = ops.pwconv(d.x1, d.a1, d.x2, d.a2, d.relu, wb, bb, 1, False)[0]
            aff = [eval_affine(bn) for bn in bns]
            rest = self.out_channels - self.n_act
            scale = torch.cat([a[0] for a in aff] + ([z.new_ones(rest)] if rest else []))
            shift = torch.cat([a[1] for a in aff] + ([z.new_zeros(rest)] if rest else []))
        tconvs = [b[3].conv for b in self.branches if not isinstance(b, nn.Conv2d) and isinstance(b[3], unit_tcn)]
        tw, tb = [c.weight for c in tconvs], [c.bias for c in tconvs]
        bn1 = self.transform[0]
        f, a1 = op_bn(bn1, lambda g_, b_, eps, want: ops.temporal_branches_bn(
            z, scale, shift, self.n_act, self.ms_cfg, self.widths, tw, tb, self.stride, g_, b_, eps, want),
            lambda f: f.shape[0] * f.shape[2] * f.shape[3])
        zt, _, a2 = conv_bn(f, a1, None, None, True, self.transform[2], 1, False, self.bn)
        return Deferred(zt, a2, None, None, False)


class MSTCN(nn.Module):
    """CTR-GCN's multi-scale temporal unit (reference: pyskl/models/gcns/utils/msg3d_utils.py:64-149): per dilation
    [1x1 -> BN -> ReLU -> (k,1) dilated conv -> BN], [1x1 -> BN -> ReLU -> max-pool(3,1) -> BN], [1x1 stride -> BN];
    cat; (+ residual); ReLU.  HIP chain: the branch 1x1 convs are ONE K-C launch (+ statistics), then branch_act ->
    tapconv (convs / max-pool / strided copy) -> plane statistics; the closing BatchNorms ride to the consumer as a
    deferred affine."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilations=[1, 2, 3, 4], residual=True,
                 act_cfg=dict(type='ReLU'), tcn_dropout=0):
        super().__init__()
        typ = act_cfg['type'] if isinstance(act_cfg, dict) else act_cfg
        if typ != 'ReLU':
            raise NotImplementedError(f'activation {typ} is not fused by the HIP kernels (ReLU only)')
        self.num_branches = len(dilations) + 2
        bc = out_channels // self.num_branches
        rem = out_channels - bc * (self.num_branches - 1)
        if type(kernel_size) == list:
            assert len(kernel_size) == len(dilations)
        else:
            kernel_size = [kernel_size] * len(dilations)
        self.stride = stride
        self.out_channels = out_channels
        self.branch_cfg = [(int(k), int(d)) for k, d in zip(kernel_size, dilations)] + [('max', 3), '1x1']
        self.widths = [bc] * (self.num_branches - 1) + [rem]
        self.branches = nn.ModuleList([
            nn.Sequential(nn.Conv2d(in_channels, bc, kernel_size=1, padding=0), nn.BatchNorm2d(bc), nn.ReLU(),
                          unit_tcn(bc, bc, kernel_size=ks, stride=stride, dilation=dilation))
            for ks, dilation in zip(kernel_size, dilations)])
        self.branches.append(nn.Sequential(
            nn.Conv2d(in_channels, bc, kernel_size=1, padding=0), nn.BatchNorm2d(bc), nn.ReLU(),
            nn.MaxPool2d(kernel_size=(3, 1), stride=(stride, 1), padding=(1, 0)), nn.BatchNorm2d(bc)))
        self.branches.append(nn.Sequential(
            nn.Conv2d(in_channels, rem, kernel_size=1, padding=0, stride=(stride, 1)), nn.BatchNorm2d(rem)))
        if not residual:
            self.residual = None
            self.residual_kind = 'none'
        elif in_channels == out_channels and stride == 1:
            self.residual = None
            self.residual_kind = 'identity'
        else:
            self.residual = unit_tcn(in_channels, out_channels, kernel_size=1, stride=stride)
            self.residual_kind = 'conv'
        self.act = nn.ReLU()
        self.drop = nn.Dropout(tcn_dropout)

    def flat_groups(self):
        """(see dgmstcn.flat_groups) the branch 1x1 convs run as one stacked conv; the BatchNorms behind them and the ones
        that close the branches as two stacked affines"""
        convs = [b[0] for b in self.branches]
        bns = [b[1] for b in self.branches[:-1]]
        post = [b[3].bn for b in self.branches[:-2]] + [self.branches[-2][4], self.branches[-1][1]]
        return [[c.weight for c in convs], [c.bias for c in convs], [bn.weight for bn in bns], [bn.bias for bn in bns],
                [bn.weight for bn in post], [bn.bias for bn in post]]

    def _branches_deferred(self, g):
        """cat of the branch outputs before the closing BatchNorms: -> (o raw, (scale, shift))."""
        ops = kernels.ops()
        d = as_deferred(g)
        n, _, T, V = d.x1.shape
        convs = [b[0] for b in self.branches]
        wb = kernels.cat_rows([c.weight.flatten(1) for c in convs])
        bb = kernels.cat_rows([c.bias for c in convs])
        bns = [b[1] for b in self.branches[:-1]]
        n_act = sum(bn.num_features for bn in bns)
        if any(_need_stats(bn) for bn in bns):
            gamma = kernels.cat_rows([bn.weight for bn in bns])
            beta = kernels.cat_rows([bn.bias for bn in bns])
            z, _, scale, shift, m, var = ops.pwconv(d.x1, d.a1, d.x2, d.a2, d.relu, wb, bb, 1, False, gamma, beta,
                                                   bns[0].eps, n_act, True)
            c0 = 0
            for bn in bns:
                record_running(bn, m[c0:c0 + bn.num_features], var[c0:c0 + bn.num_features], n * T * V)
                c0 += bn.num_features
        else:
            z = ops.pwconv(d.x1, d.a1, d.x2, d.a2, d.relu, wb, bb, 1, False)[0]
            aff = [eval_affine(bn) for bn in bns]
            rest = self.out_channels - n_act
            scale = torch.cat([a[0] for a in aff] + [z.new_ones(rest)])
            shift = torch.cat([a[1] for a in aff] + [z.new_zeros(rest)])
        tcns = [b[3] for b in self.branches[:-2]]
        tw, tb = [t.conv.weight for t in tcns], [t.conv.bias for t in tcns]
        post = [t.bn for t in tcns] + [self.branches[-2][4], self.branches[-1][1]]
        if any(_need_stats(bn) for bn in post):
            gamma = kernels.cat_rows([bn.weight for bn in post])
            beta = kernels.cat_rows([bn.bias for bn in post])
            o, s2, h2, m2, v2 = ops.temporal_branches_bn(z, scale, shift, n_act, self.branch_cfg, self.widths, tw, tb,
                                                         self.stride, gamma, beta, post[0].eps, True)
            c0 = 0
            cnt = o.shape[0] * o.shape[2] * o.shape[3]
            for bn in post:
                record_running(bn, m2[c0:c0 + bn.num_features], v2[c0:c0 + bn.num_features], cnt)
                c0 += bn.num_features
            return o, (s2, h2)
        o = ops.temporal_branches_bn(z, scale, shift, n_act, self.branch_cfg, self.widths, tw, tb, self.stride)[0]
        aff = [eval_affine(bn) for bn in post]
        return o, (torch.cat([a[0] for a in aff]), torch.cat([a[1] for a in aff]))

    def forward_deferred(self, g):
        """residual=False form (the one CTRGCNBlock builds): -> Deferred(o, closing-BN affine, relu=True)."""
        if self.residual_kind != 'none':
            raise NotImplementedError('MSTCN.forward_deferred covers residual=False; use forward()')
        o, a = self._branches_deferred(g)
        return Deferred(o, a, None, None, True)

    def forward(self, x):
        o, a = self._branches_deferred(x)
        x2 = a2 = None
        if self.residual_kind == 'identity':
            x2 = x
        elif self.residual_kind == 'conv':
            r = self.residual.forward_deferred(x)
            x2, a2 = r.x1, r.a1
        out = kernels.ops().fuse_out(o, a, x2, a2, True, False)[0]
        flush_running_stats()
        return self.drop(out) if self.drop.p > 0 else out

    def init_weights(self):
        from .gcn_units import _kaiming_conv_init
        _kaiming_conv_init(self)


class unitmlp(nn.Module):
    """Temporal "mlp" unit of the shipped CTR-GCN config (reference: tcn.py:525-614): a depthwise causal Conv1d over the
    frames of every joint (kernel (k+1)/2, left zero padding), a 1x1 conv, and — ``add_tcn`` — a dilated (k,1) conv of the
    input scaled by the learnable ``alpha``, merged after (``merge_after``) or before the 1x1 conv.  Owns the parameters;
    the arithmetic runs batched over the branches in ``msmlp``.  ``channel_annention`` raises."""

    def __init__(self, in_channels, out_channels, kernel_size=5, stride=1, dilation=1, norm='BN', dropout=0,
                 adaptive=True, channel_annention=False, reduce=4, add_tcn=False, merge_after=False):
        super().__init__()
        if channel_annention:
            raise NotImplementedError('unitmlp(channel_annention=True) has no HIP path')
        if in_channels != out_channels:
            raise NotImplementedError('unitmlp: the depthwise Conv1d needs in_channels == out_channels')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.dilation, self.stride = kernel_size, dilation, stride
        self.mlp_size = int((kernel_size + 1) / 2)
        self.channel_annention, self.add_tcn, self.merge_after = channel_annention, add_tcn, merge_after
        pad = (kernel_size + (kernel_size - 1) * (dilation - 1) - 1) // 2
        self.inter_channels = 8 if in_channels <= 16 else in_channels // reduce
        self.group = 1
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=self.mlp_size, stride=stride, dilation=dilation,
                              groups=out_channels)
        self.conv1 = nn.Conv2d(out_channels, out_channels, 1)
        if add_tcn:
            self.conv2 = nn.Conv2d(in_channels, out_channels, kernel_size=(kernel_size, 1), padding=(pad, 0),
                                   stride=(stride, 1), dilation=(dilation, 1))
            if adaptive:
                self.alpha = nn.Parameter(torch.zeros(1))
            else:
                self.register_buffer('alpha', torch.ones(1))
        self.bn = _norm_layer(norm, out_channels) if norm is not None else nn.Identity()
        self.drop = nn.Dropout(dropout, inplace=True)
        self.register_buffer('dw_dil', torch.full((out_channels,), int(dilation), dtype=torch.int32), persistent=False)

    def fusable_pairs(self):
        # conv1 feeds the BatchNorm directly only when nothing is added after it
        return [(self.conv1, self.bn)] if not (self.add_tcn and self.merge_after) else []

    def forward_deferred(self, x):
        """The unit on its own (``STGCN(tcn_type='unitmlp')``, stgcn.py:51-52 — the reference's shipped
        configs/stgcn/STGCN_model.py): x tensor or Deferred -> Deferred(out, affine of self.bn)."""
        ops = kernels.ops()
        d = as_deferred(x)
        h = d.x1 if (d.a1 is None and d.x2 is None and not d.relu) else d.materialize()
        tw = tb = None
        if self.add_tcn:                          # KB-sized products on parameters; autograd returns d alpha
            tw, tb = self.conv2.weight * self.alpha, self.conv2.bias * self.alpha
        args = (h, self.conv.weight.flatten(1), self.conv.bias, self.dw_dil, tw, tb, self.dilation,
                self.conv1.weight.flatten(1), self.conv1.bias, self.merge_after, self.stride)
        if not isinstance(self.bn, nn.BatchNorm2d):
            return Deferred(ops.temporal_unitmlp_bn(*args)[0], None, None, None, False)
        out, a = op_bn(self.bn, lambda g, b, eps, want: ops.temporal_unitmlp_bn(*args, g, b, eps, want),
                       lambda o: o.shape[0] * o.shape[2] * o.shape[3])
        return Deferred(out, a, None, None, False)

    def forward(self, x):
        out = self.forward_deferred(x).materialize()
        flush_running_stats()
        return self.drop(out) if self.drop.p > 0 else out

    def init_weights(self):
        # reference: tcn.py:611-614 (conv_init = kaiming_normal fan_out + zero bias; bn_init(bn, 1))
        for c in (self.conv, self.conv1):
            nn.init.kaiming_normal_(c.weight, mode='fan_out')
            nn.init.constant_(c.bias, 0)
        if isinstance(self.bn, nn.BatchNorm2d):
            nn.init.constant_(self.bn.weight, 1)
            nn.init.constant_(self.bn.bias, 0)


class msmlp(dgmstcn):
    """Multi-scale temporal unit of the shipped CTR-GCN config (reference: tcn.py:182-261) = ``mstcn`` with ``unitmlp``
    in place of the dilated-conv branches.  HIP chain: branch 1x1 convs as ONE K-C launch (+ statistics), then branch_act,
    tapconv (alpha-scaled dilated convs / max-pool / strided copy), the depthwise causal taps (dsgcn_dwcausal), the
    branches' 1x1 convs as one block-diagonal K-C launch, one add; transform's BN rides to its conv as a deferred affine."""

    def __init__(self, in_channels, out_channels, mid_channels=None, dropout=0.,
                 ms_cfg=[(3, 1), (3, 2), (3, 3), (3, 4), ('max', 3), '1x1'], stride=1, channel_annention=False,
                 add_tcn=False, merge_after=False):
        nn.Module.__init__(self)
        if not add_tcn:
            raise NotImplementedError('msmlp(add_tcn=False) has no HIP path (the shipped config sets add_tcn=True)')
        self.ms_cfg = [tuple(c) if isinstance(c, (list, tuple)) else c for c in ms_cfg]
        num_branches = len(ms_cfg)
        self.num_branches, self.in_channels, self.out_channels, self.stride = num_branches, in_channels, out_channels, stride
        self.act = nn.ReLU()
        self.merge_after = merge_after
        if mid_channels is None:
            mid_channels = out_channels // num_branches
            rem_mid_channels = out_channels - mid_channels * (num_branches - 1)
        else:
            assert isinstance(mid_channels, float) and mid_channels > 0
            mid_channels = int(out_channels * mid_channels)
            rem_mid_channels = mid_channels
        self.mid_channels, self.rem_mid_channels = mid_channels, rem_mid_channels
        branches, self.widths = [], []
        for i, cfg in enumerate(self.ms_cfg):
            bc = rem_mid_channels if i == 0 else mid_channels
            self.widths.append(bc)
            if cfg == '1x1':
                branches.append(nn.Conv2d(in_channels, bc, kernel_size=1, stride=(stride, 1)))
            elif cfg[0] == 'max':
                branches.append(nn.Sequential(
                    nn.Conv2d(in_channels, bc, kernel_size=1), nn.BatchNorm2d(bc), self.act,
                    nn.MaxPool2d(kernel_size=(cfg[1], 1), stride=(stride, 1), padding=(1, 0))))
            else:
                branches.append(nn.Sequential(
                    nn.Conv2d(in_channels, bc, kernel_size=1), nn.BatchNorm2d(bc), self.act,
                    unitmlp(bc, bc, kernel_size=cfg[0], stride=stride, dilation=cfg[1], norm=None,
                            channel_annention=channel_annention, add_tcn=add_tcn, merge_after=merge_after)))
        self.branches = nn.ModuleList(branches)
        tin_channels = mid_channels * (num_branches - 1) + rem_mid_channels
        self.transform = nn.Sequential(nn.BatchNorm2d(tin_channels), self.act,
                                       nn.Conv2d(tin_channels, out_channels, kernel_size=1))
        self.bn = nn.BatchNorm2d(out_channels)
        self.drop = nn.Dropout(dropout, inplace=True)
        seen_plain = False
        for cfg in self.ms_cfg:
            if cfg == '1x1':
                seen_plain = True
            elif seen_plain:
                raise NotImplementedError("msmlp: '1x1' branches must come last in ms_cfg on the HIP path")
        self.n_act = sum(w for w, c in zip(self.widths, self.ms_cfg) if c != '1x1')
        kms = {b[3].mlp_size for b in self.branches if not isinstance(b, nn.Conv2d) and isinstance(b[3], unitmlp)}
        if len(kms) != 1:
            raise NotImplementedError('msmlp: the mlp branches must share one kernel size on the HIP path')
        self.mlp_size = kms.pop()
        dil = []
        for w, cfg, b in zip(self.widths, self.ms_cfg, self.branches):
            dil += [int(cfg[1]) if (not isinstance(b, nn.Conv2d) and isinstance(b[3], unitmlp)) else 0] * w
        self.register_buffer('dw_dil', torch.tensor(dil, dtype=torch.int32), persistent=False)

    def forward_deferred(self, g):
        ops = kernels.ops()
        d = as_deferred(g)
        n, _, T, V = d.x1.shape
        convs = self._first_convs()
        wb = kernels.cat_rows([c.weight.flatten(1) for c in convs])
        bb = kernels.cat_rows([c.bias for c in convs])
        bns = [b[1] for b in self.branches if not isinstance(b, nn.Conv2d)]
        if any(_need_stats(bn) for bn in bns):
            gamma = kernels.cat_rows([bn.weight for bn in bns])
            beta = kernels.cat_rows([bn.bias for bn in bns])
            z, _, scale, shift, m, var = ops.pwconv(d.x1, d.a1, d.x2, d.a2, d.relu, wb, bb, 1, False, gamma, beta,
                                                   bns[0].eps, self.n_act, True)
            c0 = 0
            for bn in bns:
                record_running(bn, m[c0:c0 + bn.num_features], var[c0:c0 + bn.num_features], n * T * V)
                c0 += bn.num_features
        else:
            z = ops.pwconv(d.x1, d.a1, d.x2, d.a2, d.relu, wb, bb, 1, False)[0]
            aff = [eval_affine(bn) for bn in bns]
            rest = self.out_channels - self.n_act
            scale = torch.cat([a[0] for a in aff] + ([z.new_ones(rest)] if rest else []))
            shift = torch.cat([a[1] for a in aff] + ([z.new_zeros(rest)] if rest else []))
        mlps = [b[3] for b in self.branches if not isinstance(b, nn.Conv2d) and isinstance(b[3], unitmlp)]
        # the dilated convs carry their window's alpha (KB-sized products on parameters; autograd returns d alpha)
        tw = [u.conv2.weight * u.alpha for u in mlps]
        tb = [u.conv2.bias * u.alpha for u in mlps]
        C = self.out_channels
        nm = sum(u.out_channels for u in mlps)
        dw_w = torch.cat([u.conv.weight.flatten(1) for u in mlps] + [z.new_zeros(C - nm, self.mlp_size)], 0)
        dw_b = torch.cat([u.conv.bias for u in mlps] + [z.new_zeros(C - nm)], 0)
        blocks = [u.conv1.weight.flatten(1) for u in mlps]
        if self.merge_after:       # conv1(dw) + alpha*tconv: zero blocks leave max-pool / pass-through to the add
            blocks.append(z.new_zeros(C - nm, C - nm))
        else:                      # conv1(dw + alpha*tconv): identity blocks pass the other windows through the mix
            blocks.append(torch.eye(C - nm, device=z.device, dtype=z.dtype))
        pw_w = torch.block_diag(*blocks)
        pw_b = torch.cat([u.conv1.bias for u in mlps] + [z.new_zeros(C - nm)], 0)
        bn1 = self.transform[0]
        f, a1 = op_bn(bn1, lambda g_, b_, eps, want: ops.temporal_mlp_bn(
            z, scale, shift, self.n_act, self.ms_cfg, self.widths, tw, tb, dw_w, dw_b, self.dw_dil, pw_w, pw_b,
            self.merge_after, self.stride, g_, b_, eps, want), lambda f: f.shape[0] * f.shape[2] * f.shape[3])
        zt, _, a2 = conv_bn(f, a1, None, None, True, self.transform[2], 1, False, self.bn)
        return Deferred(zt, a2, None, None, False)
