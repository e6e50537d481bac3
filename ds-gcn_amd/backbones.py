"""Backbones behind the reference's registry names and ctor kwargs.

``DGSTGCN`` (reference: pyskl/models/gcns/dgstgcn.py:73-169) with ``DGBlock`` (dgstgcn.py:12-70) and
``STGCN`` (pyskl/models/gcns/stgcn.py:71-153) with ``STGCNBlock`` (stgcn.py:16-68).  state_dict keys,
block plan (channels/strides/residuals) and the ``gcn_*``/``tcn_*`` kwarg routing are the reference's; the
blocks run fused: gcn -> (deferred BN) -> tcn -> (deferred BN) -> one ``fuse_out`` that applies the last
BN, adds the residual, applies ReLU, writes the block output once and emits its time-mean for the next
block's dynamic adjacency.
"""
import copy as cp

import torch
import torch.nn as nn

from . import kernels
from .builder import BACKBONES
from .gcn_units import dggcn, dgphgcn1, unit_aagcn, unit_ctrgcn, unit_ctrhgcn, unit_gcn, flush_running_stats
from .graph import Graph
from .tcn_units import MSTCN, dgmstcn, msmlp, mstcn, unit_tcn, unitmlp

EPS = 1e-4


def _split_kwargs(kwargs, common=('act', 'norm', 'g1x1')):
    kwargs = dict(kwargs)
    for arg in common:
        if arg in kwargs:
            value = kwargs.pop(arg)
            kwargs['tcn_' + arg] = value
            kwargs['gcn_' + arg] = value
    gcn_kwargs = {k[4:]: v for k, v in kwargs.items() if k[:4] == 'gcn_'}
    tcn_kwargs = {k[4:]: v for k, v in kwargs.items() if k[:4] == 'tcn_'}
    rest = {k: v for k, v in kwargs.items() if k[:4] not in ('gcn_', 'tcn_')}
    assert len(rest) == 0, f'Invalid arguments: {rest}'
    return gcn_kwargs, tcn_kwargs


class _FusedBlock(nn.Module):
    """ReLU(tcn(gcn(x)) + residual(x)) with every BN deferred into its consumer."""

    residual_kind = 'none'   # 'none' | 'identity' | 'conv'

    def _set_residual(self, in_channels, out_channels, stride, residual):
        if not residual:
            self.residual_kind = 'none'
            self.residual = lambda x: 0
        elif in_channels == out_channels and stride == 1:
            self.residual_kind = 'identity'
            self.residual = lambda x: x
        else:
            self.residual_kind = 'conv'
            self.residual = unit_tcn(in_channels, out_channels, kernel_size=1, stride=stride)

    xbar_ld = True           # layout of the time mean handed to the next block: True = (n, C, V); an int = padded joint row

    def wants_prestrided(self):
        """True when the block residual is a 1x1 conv at stride 2 on the plain block input (dgstgcn.py:35-40): the previous
        block's fuse_out then writes the even frames as a tensor of their own (kernels._FuseOut, tee = 2)."""
        r = self.residual if self.residual_kind == 'conv' else None
        return (isinstance(r, unit_tcn) and r.kernel_size == 1 and r.stride == 2 and kernels.ops() is kernels
                and kernels.PRESTRIDED)

    def forward_fused(self, x, xbar=None, want_xbar=False, tee=False, pool=False):
        # the block input has up to three consumers (gcn main path, gcn residual operand, block residual): give each its
        # own alias so that their gradients are summed in one place instead of autograd's pairwise adds — inside the
        # previous block's fuse_out backward when that block handed over three aliases (tee), in one dsgcn_add3 otherwise
        xa, xb, xc = x if isinstance(x, tuple) else kernels.ops().tee3(x)
        g = self._gcn_deferred(xa, xbar, xb)
        x = xc
        r = hold = None
        batch = getattr(kernels.ops(), 'bn_batch', None)
        if self.residual_kind == 'conv' and batch is not None and type(self.tcn) is dgmstcn:
            # the residual conv reads only the block input: issued FIRST, its BatchNorm finalize waits (kernels.bn_batch) for
            # the transform conv's at the end of the temporal unit — one launch for the two (and one coefficient launch for
            # the two in fuse_out's backward)
            hold = batch()
            with hold.hold():
                r = self.residual.forward_deferred(x)
            t = self.tcn.forward_deferred(g, hold=hold)
        else:
            t = self.tcn.forward_deferred(g)
        drop_p = 0.0
        if getattr(self.tcn, 'drop', None) is not None and self.tcn.drop.p > 0 and self.training:
            if getattr(kernels.ops(), 'DROPOUT_FUSED', False) and self.tcn.drop.p < 1:
                drop_p = float(self.tcn.drop.p)      # applied inside fuse_out below: no mask tensor, no pass of its own
            else:
                t = type(t)(self.tcn.drop(t.materialize()), None, None, None, False)
        x2 = a2 = None
        if self.residual_kind == 'identity':
            x2 = x
        elif self.residual_kind == 'conv':
            if r is None:
                r = self.residual.forward_deferred(x)
            x2, a2 = r.x1, r.a1
        assert t.x2 is None
        # t.relu: the temporal unit ends in its own ReLU (CTR-GCN's MSTCN) -> ReLU on the first term, then add + ReLU
        kw = dict(dropout=drop_p) if drop_p > 0 else {}
        if pool:                # the last block under a pooling head: only the plane means (n, C) leave the block
            return kernels.ops().fuse_out_pool(t.x1, t.a1, x2, a2, 3 if t.relu else 1, **kw), None
        return kernels.ops().fuse_out(t.x1, t.a1, x2, a2, 3 if t.relu else 1, want_xbar and self.xbar_ld, tee, **kw)

    def forward(self, x, A=None):
        out = self.forward_fused(x)[0]
        flush_running_stats()
        return out

    def init_weights(self):
        pass


class DGBlock(_FusedBlock):
    xbar_ld = 32             # the next block's dynamic-adjacency projections run on 32-joint padded rows

    def __init__(self, in_channels, out_channels, A, edge_type, node_type, stride=1, residual=True, **kwargs):
        super().__init__()
        gcn_kwargs, tcn_kwargs = _split_kwargs(kwargs)
        tcn_type = tcn_kwargs.pop('type', 'unit_tcn')
        assert tcn_type in ['unit_tcn', 'mstcn', 'dgmstcn', 'dgmsmlp']
        if tcn_type == 'unit_tcn':
            self.tcn = unit_tcn(out_channels, out_channels, 9, stride=stride, **tcn_kwargs)
        elif tcn_type == 'dgmstcn':
            self.tcn = dgmstcn(out_channels, out_channels, stride=stride, **tcn_kwargs)
        else:
            raise NotImplementedError(f'tcn_type={tcn_type} is outside the DS-GCN hot path (SURVEY §8f)')
        gcn_type = gcn_kwargs.pop('type', 'dghgcn')
        assert gcn_type in ['dghgcn', 'dgphgcn', 'dgphgcn1', 'dggcn']
        if gcn_type == 'dggcn':                      # the original DG-STGCN unit (dgstgcn.py:42-43)
            self.gcn = dggcn(in_channels, out_channels, A, **gcn_kwargs)
        elif gcn_type == 'dgphgcn1':
            self.gcn = dgphgcn1(in_channels, out_channels, A, edge_type, node_type, **gcn_kwargs)
        else:
            raise NotImplementedError(f'gcn_type={gcn_type} is outside the DS-GCN hot path (SURVEY §8f)')
        self.relu = nn.ReLU()
        self._set_residual(in_channels, out_channels, stride, residual)

    def _gcn_deferred(self, x, xbar, x_res):
        return self.gcn.forward_deferred(x, xbar, x_res)


class STGCNBlock(_FusedBlock):

    def __init__(self, in_channels, out_channels, A, stride=1, residual=True, **kwargs):
        super().__init__()
        gcn_kwargs, tcn_kwargs = _split_kwargs(kwargs, common=())
        tcn_type = tcn_kwargs.pop('type', 'unit_tcn')
        assert tcn_type in ['unit_tcn', 'mstcn', 'unit_tcnedge', 'unitmlp', 'msmlp']
        gcn_type = gcn_kwargs.pop('type', 'unit_gcn')
        assert gcn_type in ['unit_gcn', 'unit_gcnedge']
        if gcn_type != 'unit_gcn' or tcn_type not in ('unit_tcn', 'mstcn', 'unitmlp', 'msmlp'):
            raise NotImplementedError(f'{gcn_type}/{tcn_type}: the HIP path covers unit_gcn with unit_tcn (ST-GCN), mstcn '
                                      '(ST-GCN++), unitmlp (the shipped configs/stgcn/STGCN_model.py) or msmlp')
        self.gcn = unit_gcn(in_channels, out_channels, A, **gcn_kwargs)
        if tcn_type == 'unit_tcn':
            self.tcn = unit_tcn(out_channels, out_channels, 9, stride=stride, **tcn_kwargs)
        elif tcn_type == 'mstcn':
            self.tcn = mstcn(out_channels, out_channels, stride=stride, **tcn_kwargs)
        elif tcn_type == 'unitmlp':
            self.tcn = unitmlp(out_channels, out_channels, 9, stride=stride, **tcn_kwargs)
        else:
            self.tcn = msmlp(out_channels, out_channels, stride=stride, **tcn_kwargs)
        self.relu = nn.ReLU()
        self._set_residual(in_channels, out_channels, stride, residual)

    def _gcn_deferred(self, x, xbar, x_res):
        return self.gcn.forward_deferred(x, x_res)


class CTRGCNBlock(_FusedBlock):
    """reference: pyskl/models/gcns/ctrgcn.py:9-66 — relu(tcn1(gcn1(x)) + residual(x))."""

    def __init__(self, in_channels, out_channels, A, edge_type, node_type, semantic_index=False, stride=1,
                 residual=True, kernel_size=5, dilations=[1, 2], tcn_dropout=0, **kwargs):
        super().__init__()
        gcn_kwargs = {k[4:]: v for k, v in kwargs.items() if k[:4] == 'gcn_'}
        tcn_kwargs = {k[4:]: v for k, v in kwargs.items() if k[:4] == 'tcn_'}
        kwargs = {k: v for k, v in kwargs.items() if k[:4] not in ['gcn_', 'tcn_']}
        assert len(kwargs) == 0, f'Invalid arguments: {kwargs}'
        tcn_type = tcn_kwargs.pop('type', 'mstcn')
        assert tcn_type in ['unit_tcn', 'mstcn', 'unit_tcnedge', 'unitmlp', 'msmlp']
        gcn_type = gcn_kwargs.pop('type', 'unit_ctrhgcn')
        assert gcn_type in ['unit_ctrgcn', 'unit_ctrhgcn']
        if tcn_type not in ('mstcn', 'msmlp'):
            raise NotImplementedError(f'tcn_type={tcn_type}: the HIP path covers mstcn (classic CTR-GCN) and msmlp (the '
                                      'shipped configs/ctrgcn/CTRGCN_model.py)')
        if gcn_type == 'unit_ctrgcn':
            self.gcn1 = unit_ctrgcn(in_channels, out_channels, A, **gcn_kwargs)
        else:
            self.gcn1 = unit_ctrhgcn(in_channels, out_channels, A, edge_type, node_type, semantic_index, **gcn_kwargs)
        if tcn_type == 'mstcn':
            self.tcn1 = MSTCN(out_channels, out_channels, kernel_size=kernel_size, stride=stride, dilations=dilations,
                              residual=False, tcn_dropout=tcn_dropout)
        else:
            self.tcn1 = msmlp(out_channels, out_channels, stride=stride, **tcn_kwargs)
        self.relu = nn.ReLU(inplace=True)
        self._set_residual(in_channels, out_channels, stride, residual)

    tcn = property(lambda self: self.tcn1)

    def _gcn_deferred(self, x, xbar, x_res):
        return self.gcn1.forward_deferred(x, xbar, x_res)


def _stage_kwargs(kwargs, num_stages):
    lw = [cp.deepcopy(kwargs) for _ in range(num_stages)]
    for k, v in kwargs.items():
        if isinstance(v, tuple) and len(v) == num_stages:
            for i in range(num_stages):
                lw[i][k] = v[i]
    return lw


def _stage_plan(in_channels, base_channels, num_stages, inflate_stages, down_stages, ch_ratio=2):
    """(in, out, stride, residual) of every block of a 10-stage skeleton backbone: the channel / stride schedule the four
    reference backbones share (dgstgcn.py:117-150, stgcn.py:104-128, aagcn.py:96-124, ctrgcn.py:97-108): stage 1 maps the
    input to ``base_channels`` without a residual (skipped when the widths already agree), stage i inflates the width by
    ``ch_ratio`` if i is in ``inflate_stages`` and halves the frames if i is in ``down_stages``."""
    plan = []
    if in_channels != base_channels:
        plan.append((in_channels, base_channels, 1, False))
    width, inflate_times = base_channels, 0
    for i in range(2, num_stages + 1):
        if i in inflate_stages:
            inflate_times += 1
        out = int(base_channels * ch_ratio ** inflate_times + EPS)
        plan.append((width, out, 1 + (i in down_stages), True))
        width = out
    return plan


class _SkeletonBackbone(nn.Module):
    """Shared forward: (N,M,T,V,C) -> data_bn -> (N*M,C,T,V) -> blocks -> (N,M,C',T',V).  ``forward(x, pool=True)`` (what
    ``RecognizerGCN.forward_train`` asks for under a 'GCN' pooling head) returns the (T', V) plane means (N, M, C') instead:
    the last block's activation is then never written (``kernels.fuse_out_pool``)."""
    supports_pool = True

    def _make_data_bn(self, data_bn_type, in_channels, num_person, V):
        self.data_bn_type = data_bn_type
        if data_bn_type == 'MVC':
            self.data_bn = nn.BatchNorm1d(num_person * in_channels * V)
        elif data_bn_type == 'VC':
            self.data_bn = nn.BatchNorm1d(in_channels * V)
        else:
            self.data_bn = nn.Identity()

    def _normalize_input(self, x):
        # 2.4 MB/step of host-side PyTorch plumbing (reference: dgstgcn.py:158-164)
        N, M, T, V, C = x.size()
        ops = kernels.ops()
        if self.data_bn_type in ('MVC', 'VC') and ops.data_bn_eligible(x, self.data_bn):
            return ops.data_bn(x, self.data_bn, self.data_bn_type)         # two launches, no permute copies
        x = x.permute(0, 1, 3, 4, 2).contiguous()
        if self.data_bn_type == 'MVC':
            x = self.data_bn(x.view(N, M * V * C, T))
        else:
            x = self.data_bn(x.view(N * M, V * C, T))
        return x.view(N, M, V, C, T).permute(0, 1, 3, 4, 2).contiguous().view(N * M, C, T, V)

    def _run_blocks(self, x, blocks, needs_xbar, pool=False):
        xbar = None
        last = len(blocks) - 1
        for i, blk in enumerate(blocks):
            tee = i < last
            T, V = (x[0] if isinstance(x, tuple) else x).shape[2:]
            if tee and getattr(blocks[i + 1], 'wants_prestrided', lambda: False)() and kernels.prestrided_fits(T, V):
                tee = 2                     # the next block's residual conv reads the even frames only: hand them over as a tensor
            x, xbar = blk.forward_fused(x, xbar, needs_xbar and i < last, tee=tee, pool=pool and i == last)
        flush_running_stats()
        return x

    def _shape_out(self, x, N, M):
        """(N*M, C, T, V) -> (N, M, C, T, V); plane means (N*M, C) (``forward(x, pool=True)``) -> (N, M, C)."""
        return x.reshape((N, M) + x.shape[1:])

    def init_weights(self):
        if isinstance(getattr(self, 'pretrained', None), str):
            from .checkpoint import load_checkpoint
            load_checkpoint(self, self.pretrained, strict=False)


@BACKBONES.register_module()
class DGSTGCN(_SkeletonBackbone):

    def __init__(self, graph_cfg, in_channels=3, base_channels=64, ch_ratio=2, num_stages=10, inflate_stages=[5, 8],
                 down_stages=[5, 8], data_bn_type='VC', num_person=2, pretrained=None, **kwargs):
        super().__init__()
        self.graph = Graph(**graph_cfg)
        A = torch.tensor(self.graph.A, dtype=torch.float32, requires_grad=False)
        node_type = torch.tensor(self.graph.node_type, requires_grad=False)
        edge_type = torch.tensor(self.graph.edge_type, dtype=torch.float32, requires_grad=False)
        self.kwargs = kwargs
        self._make_data_bn(data_bn_type, in_channels, num_person, A.size(1))

        lw_kwargs = _stage_kwargs(kwargs, num_stages)
        lw_kwargs[0].pop('tcn_dropout', None)
        lw_kwargs[0].pop('g1x1', None)
        lw_kwargs[0].pop('gcn_g1x1', None)
        if 'gcn_stage' in kwargs:
            for i in range(num_stages):
                lw_kwargs[i]['gcn_stage'] = i in kwargs['gcn_stage']

        self.in_channels = in_channels
        self.base_channels = base_channels
        self.ch_ratio = ch_ratio
        self.inflate_stages = inflate_stages
        self.down_stages = down_stages
        plan = _stage_plan(in_channels, base_channels, num_stages, inflate_stages, down_stages, ch_ratio)
        first = num_stages - len(plan)              # 0, or 1 when stage 1 is skipped (its kwargs slot stays unused)
        modules = [DGBlock(ci, co, A.clone(), edge_type, node_type, stride, residual=res, **lw_kwargs[first + j])
                   for j, (ci, co, stride, res) in enumerate(plan)]
        self.num_stages = len(plan)
        self.gcn = nn.ModuleList(modules)
        self.pretrained = pretrained

    def forward(self, x, pool=False):
        N, M = x.shape[:2]
        x = self._normalize_input(x)
        x = self._run_blocks(x, self.gcn[:self.num_stages], True, pool)
        return self._shape_out(x, N, M)


@BACKBONES.register_module()
class STGCN(_SkeletonBackbone):

    def __init__(self, graph_cfg, in_channels=3, base_channels=64, data_bn_type='VC', ch_ratio=2, num_person=2,
                 num_stages=10, inflate_stages=[5, 8], down_stages=[5, 8], pretrained=None, **kwargs):
        super().__init__()
        self.graph = Graph(**graph_cfg)
        A = torch.tensor(self.graph.A, dtype=torch.float32, requires_grad=False)
        self.kwargs = kwargs
        self._make_data_bn(data_bn_type, in_channels, num_person, A.size(1))
        lw_kwargs = _stage_kwargs(kwargs, num_stages)
        lw_kwargs[0].pop('tcn_dropout', None)
        self.in_channels = in_channels
        self.base_channels = base_channels
        self.ch_ratio = ch_ratio
        self.inflate_stages = inflate_stages
        self.down_stages = down_stages
        plan = _stage_plan(in_channels, base_channels, num_stages, inflate_stages, down_stages, ch_ratio)
        first = num_stages - len(plan)
        modules = [STGCNBlock(ci, co, A.clone(), stride, residual=res, **lw_kwargs[first + j])
                   for j, (ci, co, stride, res) in enumerate(plan)]
        self.num_stages = len(plan)
        self.gcn = nn.ModuleList(modules)
        self.pretrained = pretrained

    def forward(self, x, pool=False):
        N, M = x.shape[:2]
        x = self._normalize_input(x.float())
        x = self._run_blocks(x, self.gcn[:self.num_stages], False, pool)
        return self._shape_out(x, N, M)


class AAGCNBlock(_FusedBlock):
    """reference: pyskl/models/gcns/aagcn.py:12-54 — relu(tcn(gcn(x)) + residual(x))."""

    def __init__(self, in_channels, out_channels, A, edge_type, node_type, stride=1, residual=True, **kwargs):
        super().__init__()
        gcn_kwargs, tcn_kwargs = _split_kwargs(kwargs)
        tcn_type = tcn_kwargs.pop('type', 'unit_tcn')
        assert tcn_type in ['unit_tcn', 'mstcn', 'unitmlp', 'msmlp']
        gcn_type = gcn_kwargs.pop('type', 'unit_aagcn')
        assert gcn_type in ['unit_aagcn', 'unit_aahgcn']
        if gcn_type != 'unit_aagcn' or tcn_type not in ('unit_tcn', 'mstcn'):
            raise NotImplementedError(f'{gcn_type}/{tcn_type}: the HIP path covers unit_aagcn with unit_tcn or mstcn')
        self.gcn = unit_aagcn(in_channels, out_channels, A, **gcn_kwargs)
        if tcn_type == 'unit_tcn':
            self.tcn = unit_tcn(out_channels, out_channels, 9, stride=stride, **tcn_kwargs)
        else:
            self.tcn = mstcn(out_channels, out_channels, stride=stride, **tcn_kwargs)
        self.relu = nn.ReLU()
        self._set_residual(in_channels, out_channels, stride, residual)

    def _gcn_deferred(self, x, xbar, x_res):
        return self.gcn.forward_deferred(x, xbar, x_res)

    def init_weights(self):
        self.tcn.init_weights()
        self.gcn.init_weights()


@BACKBONES.register_module()
class AAGCN(_SkeletonBackbone):
    """reference: pyskl/models/gcns/aagcn.py:57-142."""

    def __init__(self, graph_cfg, in_channels=3, base_channels=64, data_bn_type='MVC', num_person=2, num_stages=10,
                 inflate_stages=[5, 8], down_stages=[5, 8], pretrained=None, **kwargs):
        super().__init__()
        self.graph = Graph(**graph_cfg)
        A = torch.tensor(self.graph.A, dtype=torch.float32, requires_grad=False)
        self.register_buffer('A', A)
        self.kwargs = kwargs
        node_type = torch.tensor(self.graph.node_type, requires_grad=False)
        edge_type = torch.tensor(self.graph.edge_type, dtype=torch.float32, requires_grad=False)
        assert data_bn_type in ['MVC', 'VC', None]
        self.in_channels = in_channels
        self.base_channels = base_channels
        self.num_person = num_person
        self.num_stages = num_stages
        self.inflate_stages = inflate_stages
        self.down_stages = down_stages
        self._make_data_bn(data_bn_type, in_channels, num_person, A.size(1))
        lw_kwargs = _stage_kwargs(kwargs, num_stages)
        lw_kwargs[0].pop('tcn_dropout', None)
        plan = _stage_plan(in_channels, base_channels, num_stages, inflate_stages, down_stages)
        first = num_stages - len(plan)
        modules = [AAGCNBlock(ci, co, A.clone(), edge_type, node_type, stride, residual=res, **lw_kwargs[first + j])
                   for j, (ci, co, stride, res) in enumerate(plan)]
        self.num_stages = len(plan)
        self.gcn = nn.ModuleList(modules)
        self.pretrained = pretrained

    def init_weights(self):
        if isinstance(self.data_bn, nn.BatchNorm1d):
            nn.init.constant_(self.data_bn.weight, 1)
            nn.init.constant_(self.data_bn.bias, 0)
        for module in self.gcn:
            module.init_weights()
        super().init_weights()

    def forward(self, x, pool=False):
        N, M = x.shape[:2]
        x = self._normalize_input(x)
        x = self._run_blocks(x, self.gcn[:self.num_stages], False, pool)
        return self._shape_out(x, N, M)


@BACKBONES.register_module()
class CTRGCN(_SkeletonBackbone):
    """reference: pyskl/models/gcns/ctrgcn.py:69-123."""

    def __init__(self, graph_cfg, in_channels=3, base_channels=64, num_stages=10, inflate_stages=[5, 8],
                 down_stages=[5, 8], semantic_stage=range(1, 11), pretrained=None, num_person=2, **kwargs):
        super().__init__()
        self.graph = Graph(**graph_cfg)
        A = torch.tensor(self.graph.A, dtype=torch.float32, requires_grad=False)
        self.register_buffer('A', A)
        node_type = torch.tensor(self.graph.node_type, requires_grad=False)
        edge_type = torch.tensor(self.graph.edge_type, dtype=torch.float32, requires_grad=False)
        self.num_person = num_person
        self.base_channels = base_channels
        self._make_data_bn('MVC', in_channels, num_person, A.size(1))
        kwargs0 = {k: v for k, v in kwargs.items() if k != 'tcn_dropout'}
        plan = [(in_channels, base_channels, 1, False)] + _stage_plan(base_channels, base_channels, num_stages,
                                                                       inflate_stages, down_stages)
        modules = [CTRGCNBlock(ci, co, A.clone(), edge_type, node_type, (j + 1) in semantic_stage, stride=stride,
                               residual=res, **(kwargs if j else kwargs0))
                   for j, (ci, co, stride, res) in enumerate(plan)]
        self.net = nn.ModuleList(modules)
        self.pretrained = pretrained

    def forward(self, x, pool=False):
        N, M = x.shape[:2]
        x = self._normalize_input(x)
        x = self._run_blocks(x, self.net, True, pool)
        return self._shape_out(x, N, M)
