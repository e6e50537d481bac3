"""CPU oracle for the DS-GCN hot path — TEST INFRASTRUCTURE, not product code.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product package (``ds-gcn_amd/``) never does, and fails loudly
when its HIP library is missing.

What it is: a plain-PyTorch (CPU, fp32 or fp64) restatement of the reference's arithmetic
for the hot path, written from the math in SURVEY.md Appendix A and functional over a
``state_dict`` with the reference's key names.  Each function cites the
reference lines it restates (paths relative to the reference root).

Parity pin: ``tests/test_oracle_golden.py`` (``*_live`` tests) checks every unit here against the
imported reference in the build container (skipped where /root/reference is absent), and
``tests/golden/*.npz`` (written by ``tests/golden/gen_golden.py`` from the imported
reference) pin it on the GPU box.  The reference itself ships no tests/golden vectors
(SURVEY.md §4), so these fixtures are the pin.
"""
from math import ceil

import numpy as np
import torch
import torch.nn.functional as F

EPS_BN = 1e-5


# ----------------------------------------------------------------------------------------
# graph constants  (pyskl/utils/graph.py:58-187)
# ----------------------------------------------------------------------------------------

NTU_PAIRS_1BASED = [(1, 2), (2, 21), (3, 21), (4, 3), (5, 21), (6, 5), (7, 6), (8, 7), (9, 21), (10, 9),
                    (11, 10), (12, 11), (13, 1), (14, 13), (15, 14), (16, 15), (17, 1), (18, 17), (19, 18),
                    (20, 19), (22, 8), (23, 8), (24, 12), (25, 12)]
COCO_PAIRS = [(15, 13), (13, 11), (16, 14), (14, 12), (11, 5), (12, 6), (9, 7), (7, 5), (10, 8), (8, 6),
              (5, 0), (6, 0), (1, 0), (3, 1), (2, 0), (4, 2)]
NTU_PARTS = [0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 0, 1, 1, 2, 2]
COCO_PARTS = [0, 0, 0, 0, 0, 1, 2, 1, 2, 1, 2, 3, 4, 3, 4, 3, 4]


def graph_constants(layout):
    """node_type (V,), edge_type (V,V) as the reference builds them (graph.py:107-144).

    edge class = rank of the signed product (p_u+1)(-1)^(p_u+1) * (p_w+1)(-1)^(p_w+1) among the
    distinct products (np.unique order = ascending).
    """
    if layout == 'nturgb+d':
        parts, V = NTU_PARTS, 25
        inward = [(i - 1, j - 1) for i, j in NTU_PAIRS_1BASED]
        center = 20
    elif layout == 'coco':
        parts, V = COCO_PARTS, 17
        inward = list(COCO_PAIRS)
        center = 0
    else:
        raise ValueError(layout)
    code = np.array([(p + 1) * (-1) ** (p + 1) for p in parts]).reshape(V, 1)
    prod = code @ code.T
    uniq = np.unique(prod)
    edge_type = np.zeros((V, V))
    for r, u in enumerate(uniq):
        edge_type[prod == u] = r
    return dict(V=V, inward=inward, center=center, node_type=np.array(parts), edge_type=edge_type)


def _norm_digraph(A):
    # graph.py:27-38: column-normalise (divide column j by its sum when > 0)
    d = A.sum(0)
    out = np.zeros_like(A)
    nz = d > 0
    out[:, nz] = A[:, nz] / d[nz]
    return out


def _edge2mat(link, V):
    A = np.zeros((V, V))
    for i, j in link:
        A[j, i] = 1
    return A


def graph_A(layout, mode, max_hop=1):
    """Static adjacency for modes 'spatial' (graph.py:174-179) and 'stgcn_spatial' (151-172)."""
    g = graph_constants(layout)
    V, inward, center = g['V'], g['inward'], g['center']
    outward = [(j, i) for i, j in inward]
    if mode == 'spatial':
        return np.stack([np.eye(V), _norm_digraph(_edge2mat(inward, V)), _norm_digraph(_edge2mat(outward, V))])
    if mode == 'stgcn_spatial':
        adj1 = np.eye(V)
        for i, j in inward:
            adj1[i, j] = 1
            adj1[j, i] = 1
        hop = np.full((V, V), np.inf)
        mats = [np.linalg.matrix_power(adj1, d) for d in range(max_hop + 1)]
        for d in range(max_hop, -1, -1):
            hop[mats[d] > 0] = d
        adj = (hop <= max_hop).astype(float)
        nadj = _norm_digraph(adj)
        out = []
        for h in range(max_hop + 1):
            close = np.zeros((V, V))
            far = np.zeros((V, V))
            for i in range(V):
                for j in range(V):
                    if hop[j, i] == h:
                        if hop[j, center] >= hop[i, center]:
                            close[j, i] = nadj[j, i]
                        else:
                            far[j, i] = nadj[j, i]
            out.append(close)
            if h > 0:
                out.append(far)
        return np.stack(out)
    raise ValueError(mode)


# ----------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------

def _conv1x1(x, w, b, stride=1):
    return F.conv2d(x, w, b, stride=(stride, 1))


def _bn(x, sd, prefix, training, eps=EPS_BN):
    """Functional BatchNorm over dim 1.  Train mode: batch stats (running buffers untouched
    here — the oracle checks outputs, running-stat updates are checked separately)."""
    w, b = sd[prefix + 'weight'], sd[prefix + 'bias']
    if training:
        return F.batch_norm(x, None, None, w, b, True, 0.0, eps)
    return F.batch_norm(x, sd[prefix + 'running_mean'].to(x.dtype), sd[prefix + 'running_var'].to(x.dtype),
                        w, b, False, 0.0, eps)


def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


# ----------------------------------------------------------------------------------------
# dgphgcn1  (pyskl/models/gcns/utils/gcn.py:2074-2372, config of configs/dsstgcn/DSSTGCN_model.py)
# ----------------------------------------------------------------------------------------

def dgphgcn1_adjacency(x, sd, node_type, edge_type, K=3, P=5, E=15, ret_parts=False):
    """Dynamic adjacency Â (n,K,mid,V,V) — SURVEY App. A.1 steps 2-8 (gcn.py:2240-2337).

    sd keys: A, alpha, beta, conv1.*, conv2.*, conv1_se.*, edge_linears.*  (conv2_se unused, Q1).
    """
    n, Ci, T, V = x.shape
    S = ceil(K / 3)
    assert S == 1 and K == 3, 'oracle restates the shipped DS-STGCN configuration (K=3)'
    A = sd['A']
    mid = sd['conv1.weight'].shape[0] // (K - S)
    nt = torch.as_tensor(np.asarray(node_type), dtype=torch.long)
    et = torch.as_tensor(np.asarray(edge_type), dtype=torch.long)
    xbar = x.mean(dim=2)                                             # (n,Ci,V)  gcn.py:2246
    w1 = sd['conv1.weight'].reshape(-1, Ci)
    w2 = sd['conv2.weight'].reshape(-1, Ci)
    a = torch.einsum('oc,ncv->nov', w1, xbar) + sd['conv1.bias'][None, :, None]
    b = torch.einsum('oc,ncv->nov', w2, xbar) + sd['conv2.bias'][None, :, None]
    a = a.reshape(n, 2, mid, V)                                      # gcn.py:2248
    b = b.reshape(n, 2, mid, V)                                      # gcn.py:2249
    wse = sd['conv1_se.weight'].reshape(mid, P, Ci)                  # channel index c*P+p (gcn.py:2256)
    bse = sd['conv1_se.bias'].reshape(mid, P)
    wsel = wse[:, nt, :]                                             # (mid,V,Ci)
    s = torch.einsum('cvi,niv->ncv', wsel, xbar) + bse[:, nt][None]  # (n,mid,V)  gcn.py:2253-2259
    x1 = torch.stack([a[:, 0], a[:, 1], s], 1)                       # (n,3,mid,V) gcn.py:2271
    x2 = torch.stack([b[:, 0], b[:, 1], s], 1)                       # Q1: conv1_se on both sides (2272)
    D0 = a[:, 0, :, :, None] - b[:, 0, :, None, :]                   # (n,mid,V,V) gcn.py:2292
    diff1 = a[:, 1, :, :, None] - b[:, 1, :, None, :]
    we = sd['edge_linears.weight'].reshape(E, mid, mid)              # out index e*mid+c (gcn.py:2280)
    be = sd['edge_linears.bias'].reshape(E, mid)
    wsel_e = we[et]                                                  # (V,V,mid,mid)
    D1 = torch.einsum('uwcd,nduw->ncuw', wsel_e, diff1) + be[et].permute(2, 0, 1)[None]  # 2279-2288
    D2 = s[:, :, :, None] - s[:, :, None, :]                         # gcn.py:2293
    D = torch.stack([D0, D1, D2], 1)                                 # (n,3,mid,V,V)
    th = torch.tanh(D)                                               # gcn.py:2298
    G = torch.einsum('nkcu,nkcw->nkuw', x1, x2)                      # gcn.py:2314
    Sm = torch.softmax(G, dim=-2)                                    # softmax over FIRST vertex idx (2174,2326)
    alpha, beta = sd['alpha'], sd['beta']
    Ahat = (A[None, :, None] + alpha[None, :, None, None, None] * th
            + beta[None, :, None, None, None] * Sm[:, :, None])      # gcn.py:2304-2337
    if ret_parts:
        return Ahat, dict(xbar=xbar, x1=x1, x2=x2, D=D, Sm=Sm)
    return Ahat


def dgphgcn1_forward(x, sd, node_type, edge_type, training=True, ret_parts=False):
    """Full spatial unit (gcn.py:2217-2365): res, pre, Â, aggregate, post, BN, +res, ReLU."""
    n, Ci, T, V = x.shape
    K = sd['A'].shape[0]
    Co = sd['post.weight'].shape[0]
    if 'down.0.weight' in sd:
        res = _bn(_conv1x1(x, sd['down.0.weight'], sd['down.0.bias']), sd, 'down.1.', training)  # 2209-2214,2221
    else:
        res = x
    Pre = F.relu(_bn(_conv1x1(x, sd['pre.0.weight'], sd['pre.0.bias']), sd, 'pre.1.', training))  # 2236
    mid = Pre.shape[1] // K
    Pre5 = Pre.reshape(n, K, mid, T, V)
    out = dgphgcn1_adjacency(x, sd, node_type, edge_type, K=K, ret_parts=ret_parts)
    Ahat = out[0] if ret_parts else out
    Y = torch.einsum('nkctu,nkcuw->nkctw', Pre5, Ahat)               # gcn.py:2350-2352
    z = _conv1x1(Y.reshape(n, K * mid, T, V), sd['post.weight'], sd['post.bias'])  # 2363-2364
    y = F.relu(_bn(z, sd, 'bn.', training) + res)                    # 2365
    if ret_parts:
        parts = dict(out[1])
        parts.update(P=Pre5, Ahat=Ahat, Y=Y)
        return y, parts
    return y


def dggcn_forward(x, sd, training=True, subset_wise=False):
    """The original DG-STGCN spatial unit with its class defaults ctr='T', ada='T', tanh / softmax (gcn.py:1445-1584)."""
    n, Ci, T, V = x.shape
    A = sd['A']
    K = A.shape[0]
    if 'down.0.weight' in sd:
        res = _bn(_conv1x1(x, sd['down.0.weight'], sd['down.0.bias']), sd, 'down.1.', training)   # gcn.py:1506-1509,1517
    else:
        res = x
    Pre = F.relu(_bn(_conv1x1(x, sd['pre.0.weight'], sd['pre.0.bias']), sd, 'pre.1.', training))  # gcn.py:1522
    mid = Pre.shape[1] // K
    Pre5 = Pre.reshape(n, K, mid, T, V)
    xbar = x.mean(dim=-2, keepdim=True)                                                            # gcn.py:1530-1531
    x1 = _conv1x1(xbar, sd['conv1.weight'], sd['conv1.bias']).reshape(n, K, mid, V)                # gcn.py:1533
    x2 = _conv1x1(xbar, sd['conv2.weight'], sd['conv2.bias']).reshape(n, K, mid, V)                # gcn.py:1534
    th = torch.tanh(x1[..., :, None] - x2[..., None, :])                                           # gcn.py:1538-1539
    G = torch.einsum('nkcv,nkcw->nkvw', x1, x2)                                                    # gcn.py:1549
    Sm = torch.softmax(G, dim=-2)                                                                  # nn.Softmax(-2), 1499
    if subset_wise:
        a, b = sd['alpha'][None, :, None, None, None], sd['beta'][None, :, None, None, None]       # gcn.py:1541-1542,1552-1553
    else:
        a, b = sd['alpha'][0], sd['beta'][0]                                                       # gcn.py:1544,1555
    Ahat = A[None, :, None] + a * th + b * Sm[:, :, None]                                          # gcn.py:1545,1556
    Y = torch.einsum('nkctv,nkcvw->nkctw', Pre5, Ahat)                                             # gcn.py:1567-1568
    z = _conv1x1(Y.reshape(n, K * mid, T, V), sd['post.weight'], sd['post.bias'])                  # gcn.py:1581-1582
    return F.relu(_bn(z, sd, 'bn.', training) + res)                                               # gcn.py:1583


def unit_aagcn_forward(x, sd, training=True, adaptive=True, attention=True):
    """2s-AGCN / AAGCN spatial unit (gcn.py:349-460; f-4)."""
    n, C, T, V = x.shape
    A = sd['A']
    S = A.shape[0]
    y = None
    for i in range(S):
        if adaptive:
            a1 = _conv1x1(x, sd[f'conv_a.{i}.weight'], sd[f'conv_a.{i}.bias'])                     # gcn.py:432
            ic = a1.shape[1]
            a1 = a1.permute(0, 3, 1, 2).reshape(n, V, ic * T)
            a2 = _conv1x1(x, sd[f'conv_b.{i}.weight'], sd[f'conv_b.{i}.bias']).reshape(n, ic * T, V)   # gcn.py:433
            adj = A[i] + torch.tanh(torch.matmul(a1, a2) / (ic * T)) * sd['alpha']                # gcn.py:434-435
        else:
            adj = A[i]                                                                            # gcn.py:441
        z = _conv1x1(torch.matmul(x.reshape(n, C * T, V), adj).reshape(n, C, T, V),
                     sd[f'conv_d.{i}.weight'], sd[f'conv_d.{i}.bias'])                            # gcn.py:436-437
        y = z if y is None else z + y
    if 'down.0.weight' in sd:
        res = _bn(_conv1x1(x, sd['down.0.weight'], sd['down.0.bias']), sd, 'down.1.', training)   # gcn.py:391-395
    else:
        res = x
    y = F.relu(_bn(y, sd, 'bn.', training) + res)                                                 # gcn.py:445
    if attention:
        pj = (sd['conv_sa.weight'].shape[-1] - 1) // 2
        se1 = torch.sigmoid(F.conv1d(y.mean(-2), sd['conv_sa.weight'], sd['conv_sa.bias'], padding=pj))   # gcn.py:449-450
        y = y * se1.unsqueeze(-2) + y
        se1 = torch.sigmoid(F.conv1d(y.mean(-1), sd['conv_ta.weight'], sd['conv_ta.bias'], padding=4))    # gcn.py:453-454
        y = y * se1.unsqueeze(-1) + y
        se = y.mean(-1).mean(-1)                                                                  # gcn.py:457
        se2 = torch.sigmoid(F.linear(F.relu(F.linear(se, sd['fc1c.weight'], sd['fc1c.bias'])),
                                     sd['fc2c.weight'], sd['fc2c.bias']))                         # gcn.py:458-459
        y = y * se2.unsqueeze(-1).unsqueeze(-1) + y
    return y


def aagcn_block_forward(x, sd, stride, residual, training=True):
    """relu(tcn(gcn(x)) + residual(x)) with unit_aagcn + unit_tcn(k=9) — aagcn.py:12-54."""
    g = unit_aagcn_forward(x, _sub(sd, 'gcn.'), training)
    t = unit_tcn_forward(g, _sub(sd, 'tcn.'), 9, stride, 1, training)
    if not residual:
        res = 0
    elif 'residual.conv.weight' in sd:
        res = unit_tcn_forward(x, _sub(sd, 'residual.'), 1, stride, 1, training)
    else:
        res = x
    return F.relu(t + res)


def aagcn_plan(in_channels=3, base_channels=64, num_stages=10, inflate_stages=(5, 8), down_stages=(5, 8)):
    """(Ci, Co, stride, residual) per block — aagcn.py:106-116."""
    plan = []
    bc = base_channels
    if in_channels != bc:
        plan.append((in_channels, bc, 1, False))
    for i in range(2, num_stages + 1):
        co = bc * (1 + (i in inflate_stages))
        plan.append((bc, co, 1 + (i in down_stages), True))
        bc = co
    return plan


def aagcn_forward(x, sd, plan, training=True):
    """AAGCN.forward (aagcn.py:127-141): data_bn over (M V C), blocks."""
    N, M, T, V, C = x.shape
    h = x.permute(0, 1, 3, 4, 2).contiguous().view(N, M * V * C, T)
    h = F.batch_norm(h, sd.get('data_bn.running_mean').clone() if not training else None,
                     sd.get('data_bn.running_var').clone() if not training else None,
                     sd['data_bn.weight'], sd['data_bn.bias'], training, 0.1, 1e-5)
    h = h.view(N, M, V, C, T).permute(0, 1, 3, 4, 2).contiguous().view(N * M, C, T, V)
    for i, (_, _, stride, residual) in enumerate(plan):
        h = aagcn_block_forward(h, _sub(sd, f'gcn.{i}.'), stride, residual, training)
    return h.reshape((N, M) + h.shape[1:])


# ----------------------------------------------------------------------------------------
# unit_tcn / dgmstcn  (pyskl/models/gcns/utils/tcn.py:10-37, 344-431)
# ----------------------------------------------------------------------------------------

def unit_tcn_forward(x, sd, kernel_size, stride=1, dilation=1, training=True, norm=True):
    """Dropout_0(BN(Conv((k,1), pad, stride, dil))) — tcn.py:10-37 (dropout p=0 only)."""
    pad = (kernel_size + (kernel_size - 1) * (dilation - 1) - 1) // 2
    y = F.conv2d(x, sd['conv.weight'], sd['conv.bias'], stride=(stride, 1), padding=(pad, 0),
                 dilation=(dilation, 1))
    if norm:
        y = _bn(y, sd, 'bn.', training)
    return y


def dgmstcn_forward(x, sd, stride=1, ms_cfg=((3, 1), (3, 2), (3, 3), (3, 4), ('max', 3), '1x1'), training=True,
                    ret_parts=False):
    """Multi-scale temporal unit with the global joint (tcn.py:407-428)."""
    n, C, T, V = x.shape
    xp = torch.cat([x, x.mean(-1, keepdim=True)], -1)                # tcn.py:409
    outs = []
    for j, cfg in enumerate(ms_cfg):
        p = f'branches.{j}.'
        if cfg == '1x1':
            outs.append(_conv1x1(xp, sd[p + 'weight'], sd[p + 'bias'], stride))       # tcn.py:383
            continue
        h = F.relu(_bn(_conv1x1(xp, sd[p + '0.weight'], sd[p + '0.bias']), sd, p + '1.', training))
        if cfg[0] == 'max':
            outs.append(F.max_pool2d(h, (cfg[1], 1), (stride, 1), (1, 0)))            # tcn.py:387-390
        else:
            outs.append(unit_tcn_forward(h, _sub(sd, p + '3.'), cfg[0], stride, cfg[1], training, norm=False))
    o = torch.cat(outs, 1)                                           # tcn.py:415
    coeff = sd['add_coeff'][:V]
    f = o[..., :V] + o[..., V, None] * coeff                        # tcn.py:416-420
    h = F.relu(_bn(f, sd, 'transform.0.', training))
    zt = _conv1x1(h, sd['transform.2.weight'], sd['transform.2.bias'])               # tcn.py:401-402,422
    y = _bn(zt, sd, 'bn.', training)                                 # tcn.py:427 (dropout p=0)
    if ret_parts:
        return y, dict(o=o, f=f, zt=zt)
    return y


def mstcn_forward(x, sd, stride=1, ms_cfg=((3, 1), (3, 2), (3, 3), (3, 4), ('max', 3), '1x1'), training=True):
    """Multi-scale temporal unit of ST-GCN++ (tcn.py:104-177): dgmstcn without the global joint."""
    outs = []
    for j, cfg in enumerate(ms_cfg):
        p = f'branches.{j}.'
        if cfg == '1x1':
            outs.append(_conv1x1(x, sd[p + 'weight'], sd[p + 'bias'], stride))        # tcn.py:137
            continue
        h = F.relu(_bn(_conv1x1(x, sd[p + '0.weight'], sd[p + '0.bias']), sd, p + '1.', training))
        if cfg[0] == 'max':
            outs.append(F.max_pool2d(h, (cfg[1], 1), (stride, 1), (1, 0)))            # tcn.py:141-145
        else:
            outs.append(unit_tcn_forward(h, _sub(sd, p + '3.'), cfg[0], stride, cfg[1], training, norm=False))
    f = torch.cat(outs, 1)                                                            # tcn.py:167
    h = F.relu(_bn(f, sd, 'transform.0.', training))
    zt = _conv1x1(h, sd['transform.2.weight'], sd['transform.2.bias'])               # tcn.py:156-157,168
    return _bn(zt, sd, 'bn.', training)                                               # tcn.py:173 (dropout p=0)


# ----------------------------------------------------------------------------------------
# DGBlock / DGSTGCN / head / loss
# ----------------------------------------------------------------------------------------

def dgblock_forward(x, sd, node_type, edge_type, stride, residual, training=True, subset_wise=False):
    """ReLU(tcn(gcn(x)) + residual(x)) — dgstgcn.py:12-65."""
    gsd = _sub(sd, 'gcn.')
    if 'edge_linears.weight' in gsd:
        g = dgphgcn1_forward(x, gsd, node_type, edge_type, training)
    else:                                                            # gcn_type='dggcn' (dgstgcn.py:42-43)
        g = dggcn_forward(x, gsd, training, subset_wise=subset_wise)
    t = dgmstcn_forward(g, _sub(sd, 'tcn.'), stride, training=training)
    if not residual:
        res = 0
    elif 'residual.conv.weight' in sd:
        res = unit_tcn_forward(x, _sub(sd, 'residual.'), 1, stride, 1, training)      # dgstgcn.py:59
    else:
        res = x
    return F.relu(t + res)


def dgstgcn_plan(in_channels=3, base_channels=64, ch_ratio=2, num_stages=10, inflate_stages=(5, 8),
                 down_stages=(5, 8)):
    """(Ci, Co, stride, residual) per block — dgstgcn.py:121-142."""
    plan = []
    bc = base_channels
    if in_channels != bc:
        plan.append((in_channels, bc, 1, False))
    inflate = 0
    for i in range(2, num_stages + 1):
        stride = 1 + (i in down_stages)
        ci = bc
        if i in inflate_stages:
            inflate += 1
        co = int(base_channels * ch_ratio ** inflate + 1e-4)
        bc = co
        plan.append((ci, co, stride, True))
    return plan


def dgstgcn_forward(x, sd, node_type, edge_type, plan, training=True):
    """Backbone: (N,M,T,V,C) -> (N,M,C_out,T_out,V) — dgstgcn.py:156-170 (data_bn_type='VC')."""
    N, M, T, V, C = x.shape
    h = x.permute(0, 1, 3, 4, 2).contiguous().view(N * M, V * C, T)
    h = _bn(h, sd, 'data_bn.', training)
    h = h.view(N, M, V, C, T).permute(0, 1, 3, 4, 2).contiguous().view(N * M, C, T, V)
    for i, (ci, co, stride, residual) in enumerate(plan):
        h = dgblock_forward(h, _sub(sd, f'gcn.{i}.'), node_type, edge_type, stride, residual, training)
    return h.reshape((N, M) + h.shape[1:])


def gcn_head_forward(feat, sd):
    """GCNHead: mean over (T,V), mean over M, Linear — heads/simple_head.py:83-97 (dropout 0)."""
    N, M, C, T, V = feat.shape
    p = feat.reshape(N * M, C, T * V).mean(-1).reshape(N, M, C).mean(1)
    return F.linear(p, sd['fc_cls.weight'], sd['fc_cls.bias'])


def top_k_accuracy(scores, labels, topk=(1,)):
    """core/evaluation.py:107-126."""
    res = []
    labels = np.array(labels)[:, np.newaxis]
    for k in topk:
        pred = np.argsort(scores, axis=1)[:, -k:][:, ::-1]
        res.append(np.logical_or.reduce(pred == labels, axis=1).sum() / labels.shape[0])
    return res


def recognizer_forward_train(keypoint, label, sd, node_type, edge_type, plan, training=True):
    """RecognizerGCN.forward_train (recognizers/recognizergcn.py:20-51) + BaseHead.loss
    (heads/base.py:50-84) + CrossEntropyLoss hard-label branch (losses/cross_entropy_loss.py:75-82).
    keypoint (N,1,M,T,V,C); label (N,1).  Returns (logits, loss)."""
    assert keypoint.shape[1] == 1
    feat = dgstgcn_forward(keypoint[:, 0], _sub(sd, 'backbone.'), node_type, edge_type, plan, training)
    logits = gcn_head_forward(feat, _sub(sd, 'cls_head.'))
    loss = F.cross_entropy(logits, label.squeeze(-1))
    return logits, loss


# ----------------------------------------------------------------------------------------
# ST-GCN units (gcn.py:22-97, tcn.py:10-37, stgcn.py:16-68)   SURVEY App. A.3
# ----------------------------------------------------------------------------------------

def unit_gcn_forward(x, sd, training=True, with_res=False, adaptive='init', conv_pos='pre'):
    """ST-GCN spatial unit (gcn.py:73-97).  adaptive 'offset' / 'importance' combine A with the PA parameter
    (gcn.py:80-84); conv_pos 'post' aggregates the input per subset first and mixes the K*Ci stacked channels (89-92)."""
    n, Ci, T, V = x.shape
    A = sd['A']
    if adaptive == 'offset':
        A = A + sd['PA']                                                              # gcn.py:83
    elif adaptive == 'importance':
        A = A * sd['PA']
    K = A.shape[0]
    if conv_pos == 'pre':
        h = _conv1x1(x, sd['conv.weight'], sd['conv.bias']).view(n, K, -1, T, V)      # gcn.py:86-87
        y = torch.einsum('nkctv,kvw->nctw', h, A)                                     # gcn.py:88
    else:
        h = torch.einsum('nctv,kvw->nkctw', x, A).reshape(n, K * Ci, T, V)            # gcn.py:90-91
        y = _conv1x1(h, sd['conv.weight'], sd['conv.bias'])                           # gcn.py:92
    y = _bn(y, sd, 'bn.', training)
    if with_res:                                                                      # gcn.py:57-66,70,94
        if 'down.0.weight' in sd:
            y = y + _bn(_conv1x1(x, sd['down.0.weight'], sd['down.0.bias']), sd, 'down.1.', training)
        else:
            y = y + x
    return F.relu(y)                                                                  # gcn.py:94


def stgcn_block_forward(x, sd, stride, residual, training=True, with_res=False, tcn_type='unit_tcn', merge_after=True):
    g = unit_gcn_forward(x, _sub(sd, 'gcn.'), training, with_res)
    if tcn_type == 'mstcn':
        t = mstcn_forward(g, _sub(sd, 'tcn.'), stride, training=training)             # stgcn.py:47-48 (ST-GCN++)
    elif tcn_type == 'unitmlp':                                                       # stgcn.py:51-52 (shipped STGCN_model.py)
        ts = _sub(sd, 'tcn.')
        t = _bn(unitmlp_forward(g, ts, 9, stride, 1, 'conv2.weight' in ts, merge_after), ts, 'bn.', training)   # tcn.py:609
    else:
        t = unit_tcn_forward(g, _sub(sd, 'tcn.'), 9, stride, 1, training)             # stgcn.py:45-46 (p=0)
    if not residual:
        res = 0
    elif 'residual.conv.weight' in sd:
        res = unit_tcn_forward(x, _sub(sd, 'residual.'), 1, stride, 1, training)
    else:
        res = x
    return F.relu(t + res)


def stgcn_forward(x, sd, plan, training=True, with_res=False, tcn_type='unit_tcn', merge_after=True):
    N, M, T, V, C = x.shape
    h = x.permute(0, 1, 3, 4, 2).contiguous().view(N * M, V * C, T)
    h = _bn(h, sd, 'data_bn.', training)
    h = h.view(N, M, V, C, T).permute(0, 1, 3, 4, 2).contiguous().view(N * M, C, T, V)
    for i, (ci, co, stride, residual) in enumerate(plan):
        h = stgcn_block_forward(h, _sub(sd, f'gcn.{i}.'), stride, residual, training, with_res, tcn_type, merge_after)
    return h.reshape((N, M) + h.shape[1:])


# ----------------------------------------------------------------------------------------
# CTR-GCN units (gcn.py:634-666, 882-929; msg3d_utils.py:64-149; ctrgcn.py)   SURVEY App. A.4
# ----------------------------------------------------------------------------------------

def ctrgc_forward(x, sd, A_i, alpha):
    x1 = _conv1x1(x, sd['conv1.weight'], sd['conv1.bias']).mean(-2)                   # (n,R,V) gcn.py:652
    x2 = _conv1x1(x, sd['conv2.weight'], sd['conv2.bias']).mean(-2)
    x3 = _conv1x1(x, sd['conv3.weight'], sd['conv3.bias'])
    d = torch.tanh(x1[:, :, :, None] - x2[:, :, None, :])                             # gcn.py:655
    ah = _conv1x1(d, sd['conv4.weight'], sd['conv4.bias']) * alpha + A_i[None, None]  # gcn.py:657
    return torch.einsum('ncuv,nctu->nctv', ah, x3)                                    # gcn.py:658


def unit_ctrgcn_forward(x, sd, training=True):
    A = sd['A']
    y = None
    for i in range(A.shape[0]):
        z = ctrgc_forward(x, _sub(sd, f'convs.{i}.'), A[i], sd['alpha'])
        y = z if y is None else z + y                                                 # gcn.py:915-917
    y = _bn(y, sd, 'bn.', training)
    if 'down.0.weight' in sd:
        y = y + _bn(_conv1x1(x, sd['down.0.weight'], sd['down.0.bias']), sd, 'down.1.', training)
    else:
        y = y + x
    return F.relu(y)


def unitmlp_forward(x, sd, kernel_size, stride, dilation, add_tcn=True, merge_after=True):
    """unitmlp with norm=None (tcn.py:578-612): depthwise causal Conv1d over the frames of every joint (left zero pad
    (m-1)*d, m = (k+1)/2 taps, groups = channels), 1x1 conv, + alpha * dilated (k,1) conv of the input."""
    B, C, T, V = x.shape
    m = int((kernel_size + 1) / 2)
    xs = x.permute(0, 3, 1, 2).reshape(B * V, C, T)                                         # tcn.py:584
    xs = F.pad(xs, ((m + (m - 1) * (dilation - 1) - 1), 0))                                 # tcn.py:585-586
    y = F.conv1d(xs, sd['conv.weight'], sd['conv.bias'], stride=stride, dilation=dilation, groups=C)   # 588
    y = y.reshape(B, V, C, 1, -1).mean(-2).permute(0, 2, 3, 1)                              # tcn.py:589 (group 1)
    if add_tcn:
        pad = (kernel_size + (kernel_size - 1) * (dilation - 1) - 1) // 2
        t = F.conv2d(x, sd['conv2.weight'], sd['conv2.bias'], stride=(stride, 1), padding=(pad, 0), dilation=(dilation, 1))
        if merge_after:
            return _conv1x1(y, sd['conv1.weight'], sd['conv1.bias']) + sd['alpha'] * t     # tcn.py:594-596
        return _conv1x1(y + sd['alpha'] * t, sd['conv1.weight'], sd['conv1.bias'])          # tcn.py:599-600
    return _conv1x1(y, sd['conv1.weight'], sd['conv1.bias'])


def msmlp_forward(x, sd, stride=1, ms_cfg=((3, 1), (3, 2), (3, 3), (3, 4), ('max', 3), '1x1'), training=True, add_tcn=True,
                  merge_after=True):
    """msmlp (tcn.py:182-261): mstcn's scaffold with unitmlp in the dilated branches; cat -> BN -> ReLU -> 1x1 -> BN."""
    outs = []
    for i, cfg in enumerate(ms_cfg):
        p = f'branches.{i}.'
        if cfg == '1x1':
            outs.append(_conv1x1(x, sd[p + 'weight'], sd[p + 'bias'], stride))
            continue
        h = F.relu(_bn(_conv1x1(x, sd[p + '0.weight'], sd[p + '0.bias']), sd, p + '1.', training))
        if cfg[0] == 'max':
            outs.append(F.max_pool2d(h, (cfg[1], 1), (stride, 1), (1, 0)))
        else:
            outs.append(unitmlp_forward(h, _sub(sd, p + '3.'), cfg[0], stride, cfg[1], add_tcn, merge_after))
    feat = torch.cat(outs, 1)
    feat = F.relu(_bn(feat, sd, 'transform.0.', training))
    feat = _conv1x1(feat, sd['transform.2.weight'], sd['transform.2.bias'])
    return _bn(feat, sd, 'bn.', training)


def unit_ctrhgcn_forward(x, sd, edge_type, training=True):
    """Heterogeneous CTR unit with the shipped config's flags (gcn.py:773-880 over CTRHGC 668-771; node attention is
    off in every subset and edge attention on in subset 0 only, through the constructor's per-subset overrides 801-842).
    Per subset i: x1 = conv1(x).mean(T), x2 = conv2(x).mean(T), d = tanh(x1[u] - x2[v]); subset 0: the edge-typed conv
    gives E variants of d and pair (u,v) keeps variant eps(u,v) (737-745); S = conv4(.); topology = S*alpha_i + A_i +
    beta_i * x1^T x2 (752-760); out_i = einsum('ncuv,nctu->nctv', topology, conv3(x)) (769).  Sum, BN, + down(x), ReLU."""
    n, Ci, T, V = x.shape
    et = torch.as_tensor(np.asarray(edge_type), dtype=torch.long).reshape(-1)
    K = sd['A'].shape[0]
    y = 0
    for i in range(K):
        p = f'convs.{i}.'
        x1 = _conv1x1(x, sd[p + 'conv1.weight'], sd[p + 'conv1.bias']).mean(-2)          # gcn.py:732 (n,R,V)
        x2 = _conv1x1(x, sd[p + 'conv2.weight'], sd[p + 'conv2.bias']).mean(-2)
        x3 = _conv1x1(x, sd[p + 'conv3.weight'], sd[p + 'conv3.bias'])
        d = torch.tanh(x1.unsqueeze(-1) - x2.unsqueeze(-2))                              # gcn.py:735
        R = d.shape[1]
        if p + 'edge_att_conv.weight' in sd:                                             # gcn.py:736-745
            full = _conv1x1(d, sd[p + 'edge_att_conv.weight'], sd[p + 'edge_att_conv.bias'])   # (n, E*R, V, V)
            E = full.shape[1] // R
            full = full.view(n, E, R, V * V)
            d = torch.gather(full, 1, et.view(1, 1, 1, V * V).expand(n, 1, R, V * V))[:, 0].view(n, R, V, V)
        S = _conv1x1(d, sd[p + 'conv4.weight'], sd[p + 'conv4.bias'])                    # gcn.py:747 / 751
        topo = S * sd['alpha'][i] + sd['A'][i][None, None]                               # gcn.py:755
        if p + 'beta' in sd:
            topo = topo + torch.einsum('ncv,ncw->nvw', x1, x2)[:, None] * sd[p + 'beta']  # gcn.py:759-760
        y = y + torch.einsum('ncuv,nctu->nctv', topo, x3)                                # gcn.py:769
    y = _bn(y, sd, 'bn.', training)
    if 'down.0.weight' in sd:
        y = y + _bn(_conv1x1(x, sd['down.0.weight'], sd['down.0.bias']), sd, 'down.1.', training)
    else:
        y = y + x
    return F.relu(y)


def mstcn_msg3d_forward(x, sd, stride=1, kernel_size=5, dilations=(1, 2), training=True):
    """MSTCN of msg3d_utils.py:64-149 with residual=False (ctrgcn.py:41-48)."""
    outs = []
    nb = len(dilations) + 2
    for j, d in enumerate(dilations):
        p = f'branches.{j}.'
        h = F.relu(_bn(_conv1x1(x, sd[p + '0.weight'], sd[p + '0.bias']), sd, p + '1.', training))
        outs.append(unit_tcn_forward(h, _sub(sd, p + '3.'), kernel_size, stride, d, training))
    p = f'branches.{nb - 2}.'
    h = F.relu(_bn(_conv1x1(x, sd[p + '0.weight'], sd[p + '0.bias']), sd, p + '1.', training))
    h = F.max_pool2d(h, (3, 1), (stride, 1), (1, 0))
    outs.append(_bn(h, sd, p + '4.', training))
    p = f'branches.{nb - 1}.'
    outs.append(_bn(_conv1x1(x, sd[p + '0.weight'], sd[p + '0.bias'], stride), sd, p + '1.', training))
    return F.relu(torch.cat(outs, 1))


def ctrgcn_plan(in_channels=3, base_channels=64, num_stages=10, inflate_stages=(5, 8), down_stages=(5, 8)):
    """(Ci, Co, stride, residual) per block — ctrgcn.py:98-108."""
    plan = [(in_channels, base_channels, 1, False)]
    bc = base_channels
    for i in range(2, num_stages + 1):
        co = bc * (1 + (i in inflate_stages))
        plan.append((bc, co, 1 + (i in down_stages), True))
        bc = co
    return plan


def ctrgcn_block_forward(x, sd, stride, residual, training=True, edge_type=None):
    """CTRGCNBlock.forward (ctrgcn.py:59-61): relu(tcn1(gcn1(x)) + residual(x)).  Classic: unit_ctrgcn + MSTCN (kernel 5,
    dilations (1,2)); with edge_type: the shipped configs/ctrgcn/CTRGCN_model.py variant, unit_ctrhgcn + msmlp
    (add_tcn, merge_after)."""
    if edge_type is None:
        g = unit_ctrgcn_forward(x, _sub(sd, 'gcn1.'), training)
        t = mstcn_msg3d_forward(g, _sub(sd, 'tcn1.'), stride, 5, (1, 2), training)
    else:
        g = unit_ctrhgcn_forward(x, _sub(sd, 'gcn1.'), edge_type, training)
        t = msmlp_forward(g, _sub(sd, 'tcn1.'), stride, training=training, add_tcn=True, merge_after=True)
    if not residual:
        res = 0
    elif 'residual.conv.weight' in sd:
        res = unit_tcn_forward(x, _sub(sd, 'residual.'), 1, stride, 1, training)
    else:
        res = x
    return F.relu(t + res)


def ctrgcn_forward(x, sd, plan, training=True, edge_type=None):
    """CTRGCN.forward (ctrgcn.py:113-123): data_bn over M*V*C channels."""
    N, M, T, V, C = x.shape
    h = x.permute(0, 1, 3, 4, 2).contiguous().view(N, M * V * C, T)
    h = _bn(h, sd, 'data_bn.', training)
    h = h.view(N, M, V, C, T).permute(0, 1, 3, 4, 2).contiguous().view(N * M, C, T, V)
    for i, (ci, co, stride, residual) in enumerate(plan):
        h = ctrgcn_block_forward(h, _sub(sd, f'net.{i}.'), stride, residual, training, edge_type)
    return h.reshape((N, M) + h.shape[1:])


def recognizer_forward_train_backbone(backbone, keypoint, label, sd, plan, training=True):
    """forward_train with the ST-GCN ('stgcn'), shipped ST-GCN ('stgcn_shipped': unitmlp temporal unit), ST-GCN++ ('stgcnpp'), classic CTR-GCN ('ctrgcn'), shipped-config CTR-GCN
    ('ctrgcn_shipped': unit_ctrhgcn + msmlp, NTU graph) or AAGCN ('aagcn') backbone -> (logits, loss)."""
    assert keypoint.shape[1] == 1
    if backbone == 'ctrgcn_shipped':
        feat = ctrgcn_forward(keypoint[:, 0], _sub(sd, 'backbone.'), plan, training,
                              graph_constants('nturgb+d')['edge_type'])
    elif backbone == 'stgcn_shipped':  # configs/stgcn/STGCN_model.py: gcn_adaptive='init', tcn_type='unitmlp' (add_tcn, merge_after)
        feat = stgcn_forward(keypoint[:, 0], _sub(sd, 'backbone.'), plan, training, False, 'unitmlp', True)
    elif backbone == 'stgcnpp':        # ST-GCN++ (configs/stgcn++): gcn_adaptive='init', gcn_with_res=True, tcn_type='mstcn'
        feat = stgcn_forward(keypoint[:, 0], _sub(sd, 'backbone.'), plan, training, True, 'mstcn')
    else:
        fwd = {'stgcn': stgcn_forward, 'ctrgcn': ctrgcn_forward, 'aagcn': aagcn_forward}[backbone]
        feat = fwd(keypoint[:, 0], _sub(sd, 'backbone.'), plan, training)
    logits = gcn_head_forward(feat, _sub(sd, 'cls_head.'))
    return logits, F.cross_entropy(logits, label.squeeze(-1))
