"""Plain-PyTorch statement of every fused op the HIP library implements (test infrastructure).

Same call signatures as ``dsgcn_amd.kernels`` so that (a) each HIP op is checked against the
matching function here on identical inputs (``-m gpu`` tests, fp32 and fp64 references) and
(b) the host-side composition of ops (deferred BatchNorm wiring, state_dict mapping) can be
checked against the oracle on CPU by injecting this namespace with ``kernels.use_ops``.
Autograd of these functions is PyTorch's own.
"""
import torch
import torch.nn.functional as F

NAME = 'torch_ops(test reference)'


def _bc(p):
    return p[None, :, None, None]


def virt(x1, a1, x2, a2, relu):
    """Virtual input of a fused op: relu?( x1*s1+h1 [+ x2*s2+h2 | + x2] )."""
    v = x1 if a1 is None else x1 * _bc(a1[0]) + _bc(a1[1])
    if x2 is not None:
        v = v + (x2 if a2 is None else x2 * _bc(a2[0]) + _bc(a2[1]))
    return F.relu(v) if relu else v


def pwconv(x1, a1, x2, a2, relu, weight, bias, stride=1, aug=False, gamma=None, beta=None, eps=1e-5, n_affine=None,
           want_bn=False):
    """1x1 channel mix over the virtual input (rows subsampled by ``stride``), optionally followed by the
    train-mode BN of its output expressed as a deferred affine.

    Returns (z, zaug, scale, shift, mean, var): z (n,Co,T',V); zaug (n,Co,T') = mean_v z when ``aug`` (the
    dgmstcn global joint, tcn.py:409, by linearity of the 1x1 conv).  With ``want_bn``: mean/var are the biased
    batch statistics of z over (n,T',V) (including the zaug column when ``aug``), scale = gamma*rsqrt(var+eps),
    shift = beta - mean*scale for channels < n_affine and (1, 0) for the rest."""
    v = virt(x1, a1, x2, a2, relu)
    if stride != 1:
        v = v[:, :, ::stride]
    w2 = weight.reshape(weight.shape[0], -1)
    z = torch.einsum('oc,nctv->notv', w2, v)
    if bias is not None:
        z = z + _bc(bias)
    zaug = z.mean(-1) if aug else None
    if not want_bn:
        return z, zaug, None, None, None, None
    Co = z.shape[1]
    if n_affine is None:
        n_affine = Co if gamma is not None else 0
    full = torch.cat([z, zaug[..., None]], -1) if aug else z
    mean = full.mean((0, 2, 3))
    var = full.var((0, 2, 3), unbiased=False)
    g = gamma if gamma is not None else z.new_ones(n_affine)
    b = beta if beta is not None else z.new_zeros(n_affine)
    sc = g * torch.rsqrt(var[:n_affine] + eps)
    sh = b - mean[:n_affine] * sc
    if n_affine < Co:
        sc = torch.cat([sc, z.new_ones(Co - n_affine)])
        sh = torch.cat([sh, z.new_zeros(Co - n_affine)])
    return z, zaug, sc, sh, mean.detach(), var.detach()


def _bn_of(t, gamma, beta, eps, want_bn):
    """(t, scale, shift, mean, var): the train-mode BN of t as a deferred affine (or Nones)."""
    if not want_bn:
        return t, None, None, None, None
    mean = t.mean((0, 2, 3))
    var = t.var((0, 2, 3), unbiased=False)
    g = gamma if gamma is not None else t.new_ones(t.shape[1])
    b = beta if beta is not None else t.new_zeros(t.shape[1])
    sc = g * torch.rsqrt(var + eps)
    return t, sc, b - mean * sc, mean.detach(), var.detach()


def aggregate(zp, ap, relu, ahat):
    """y[n,c,t,w] = sum_u P[n,c,t,u] * ahat[n,c,u,w],  P = virt(zp, ap)."""
    p = virt(zp, ap, None, None, relu)
    return torch.einsum('nctu,ncuw->nctw', p, ahat)


def aggregate_sum(p, adj, K, gamma=None, beta=None, eps=1e-5, want_bn=False, per_sample=False):
    """y[n,c,t,w] = sum_k sum_u p[n,k*Co+c,t,u] * adj_k[u,w]; adj (K,V,V) shared (ST-GCN, gcn.py:81-85),
    (n,K*Co,V,V) per sample and channel (CTR-GCN, gcn.py:658 + the sum over subsets gcn.py:917-919) or, per_sample,
    (n,K,V,V) per sample shared by the channels (AAGCN, gcn.py:431-437); + BN of y."""
    n, KC, T, V = p.shape
    p5 = p.view(n, K, KC // K, T, V)
    if adj.dim() == 3:
        y = torch.einsum('nkctv,kvw->nctw', p5, adj)
    elif per_sample:
        y = torch.einsum('nkctu,nkuw->nctw', p5, adj)
    else:
        y = torch.einsum('nkctu,nkcuw->nctw', p5, adj.view(n, K, KC // K, V, V))
    return _bn_of(y, gamma, beta, eps, want_bn)


def gram(a, b):
    """G[n,u,w] = sum_{c,t} a[n,c,t,u] * b[n,c,t,w]  (AAGCN's embedding product, gcn.py:432-434 before the scaling)."""
    return torch.einsum('nctu,nctw->nuw', a, b)


def gate(y, g, mode, rmode):
    """AAGCN attention gate (gcn.py:447-459): out = y * (1 + g) with g (n,V) / (n,T) / (n,C) for mode 0 / 1 / 2; the mean
    the next gate needs: rmode 1 -> over joints (n,C,T), 2 -> over frames and joints (n,C), 0 -> None."""
    gb = g[:, None, None, :] if mode == 0 else (g[:, None, :, None] if mode == 1 else g[:, :, None, None])
    out = y * (1 + gb)
    r = out.mean(-1) if rmode == 1 else (out.mean((-1, -2)) if rmode == 2 else None)
    return out, r


def ctr_topology(xbar, w1, b1, w2, b2, w4, b4, alpha, A, beta=None, edge=None, subset_major=False):
    """CTR-GCN refined topology (gcn.py:651-657) and its CTRHGC form (gcn.py:719-760: per-subset alpha, edge-typed
    attention on chosen subsets, Gram term scaled by beta): -> Ahat (n, K*Co, V, V)."""
    n, Ci, V = xbar.shape
    K = A.shape[0]
    R = w1.shape[0] // K
    x1 = (torch.einsum('oc,ncv->nov', w1, xbar) + b1[None, :, None]).view(n, K, R, V)
    x2 = (torch.einsum('oc,ncv->nov', w2, xbar) + b2[None, :, None]).view(n, K, R, V)
    d = torch.tanh(x1[..., :, None] - x2[..., None, :])                  # n,K,R,V,V
    out = []
    for k in range(K):
        dk = d[:, k]
        if edge and k in edge:
            we, be, et = edge[k]
            E = we.shape[0] // R
            full = (torch.einsum('or,nruv->nouv', we, dk) + be[None, :, None, None]).view(n, E, R, V * V)
            dk = torch.gather(full, 1, et.long().view(1, 1, 1, V * V).expand(n, 1, R, V * V))[:, 0].view(n, R, V, V)
        s = torch.einsum('or,nruv->nouv', w4[k], dk) + b4[k][None, :, None, None]
        al = alpha[k] if alpha.numel() == K and K > 1 else alpha
        a = s * al + A[k][None, None]
        if beta is not None:
            a = a + beta[k] * torch.einsum('nru,nrv->nuv', x1[:, k], x2[:, k])[:, None]
        out.append(a)
    return torch.cat(out, 1)


def tee3(x):
    return x, x, x


def tmean(x, ld=True):
    xbar = x.mean(2)
    if ld is not True and int(ld) > xbar.shape[-1]:
        xbar = F.pad(xbar, (0, int(ld) - xbar.shape[-1]))
    return xbar


def dynadj(xbar, A, alpha, beta, w1, b1, w2, b2, wse, bse, we, be, node_type, edge_type, single_use=True):
    """Dynamic adjacency.  xbar (n,Ci,V) -> ahat (n,K*mid,V,V), K=3.

    w1,w2: (2*mid,Ci); wse: (mid*P,Ci) channel index c*P+p; we: (E*mid,mid) out index e*mid+c.
    node_type (V,) long, edge_type (V,V) long."""
    V = A.shape[-1]
    xbar = xbar[..., :V]                                   # (a padded joint row from fuse_out(want_tmean=32))
    n, Ci, _ = xbar.shape
    node_type, edge_type = node_type.long(), edge_type.long()
    K = A.shape[0]
    mid = w1.shape[0] // 2
    P = wse.shape[0] // mid
    E = we.shape[0] // mid
    a = (torch.einsum('oc,ncv->nov', w1, xbar) + b1[None, :, None]).reshape(n, 2, mid, V)
    b = (torch.einsum('oc,ncv->nov', w2, xbar) + b2[None, :, None]).reshape(n, 2, mid, V)
    wsel = wse.reshape(mid, P, Ci)[:, node_type, :]
    s = torch.einsum('cvi,niv->ncv', wsel, xbar) + bse.reshape(mid, P)[:, node_type][None]
    x1 = torch.stack([a[:, 0], a[:, 1], s], 1)
    x2 = torch.stack([b[:, 0], b[:, 1], s], 1)
    d0 = a[:, 0, :, :, None] - b[:, 0, :, None, :]
    diff1 = a[:, 1, :, :, None] - b[:, 1, :, None, :]
    wsel_e = we.reshape(E, mid, mid)[edge_type]
    d1 = torch.einsum('uwcd,nduw->ncuw', wsel_e, diff1) + be.reshape(E, mid)[edge_type].permute(2, 0, 1)[None]
    d2 = s[:, :, :, None] - s[:, :, None, :]
    th = torch.tanh(torch.stack([d0, d1, d2], 1))
    sm = torch.softmax(torch.einsum('nkcu,nkcw->nkuw', x1, x2), dim=-2)
    ahat = (A[None, :, None] + alpha[None, :, None, None, None] * th + beta[None, :, None, None, None] * sm[:, :, None])
    return ahat.reshape(n, K * mid, V, V)


def temporal_ms(z, zaug, scale, shift, n_act, branch_cfg, widths, conv_w, conv_b, add_coeff, stride, gamma=None,
                beta=None, eps=1e-5, want_bn=False):
    """Multi-scale temporal stage of dgmstcn after the fused branch 1x1 conv.

    z (n,C,T,V), zaug (n,C,T): raw branch-conv outputs (real joints / global joint).
    scale/shift (C): deferred BN affine; channels < n_act also get ReLU, the rest pass through.
    branch_cfg: list of (k,dil) | ('max',k) | '1x1'; widths: channels per branch;
    conv_w[i] (bc,bc,k,1), conv_b[i] (bc) for the conv branches in order.
    Returns (f, scale, shift, mean, var): f (n,C,T/stride,V) = local + global*add_coeff and, with ``want_bn``, the
    train-mode BN of f (gamma/beta = transform.0) as a deferred affine."""
    n, C, T, V = z.shape
    full = torch.cat([z, zaug[..., None]], -1)
    h = full * _bc(scale) + _bc(shift)
    h = torch.cat([F.relu(h[:, :n_act]), h[:, n_act:]], 1)
    outs = []
    c0 = 0
    ci = 0
    for cfg, bc in zip(branch_cfg, widths):
        hb = h[:, c0:c0 + bc]
        if cfg == '1x1':
            outs.append(hb[:, :, ::stride])
        elif cfg[0] == 'max':
            outs.append(F.max_pool2d(hb, (cfg[1], 1), (stride, 1), (1, 0)))
        else:
            k, d = cfg
            pad = (k + (k - 1) * (d - 1) - 1) // 2
            outs.append(F.conv2d(hb, conv_w[ci], conv_b[ci], stride=(stride, 1), padding=(pad, 0), dilation=(d, 1)))
            ci += 1
        c0 += bc
    o = torch.cat(outs, 1)
    f = o[..., :V] + o[..., V, None] * add_coeff[:V]
    if not want_bn:
        return f, None, None, None, None
    mean = f.mean((0, 2, 3))
    var = f.var((0, 2, 3), unbiased=False)
    g = gamma if gamma is not None else f.new_ones(C)
    b = beta if beta is not None else f.new_zeros(C)
    sc = g * torch.rsqrt(var + eps)
    return f, sc, b - mean * sc, mean.detach(), var.detach()


def _branches(h, branch_cfg, widths, conv_w, conv_b, stride):
    outs = []
    c0 = ci = 0
    for cfg, bc in zip(branch_cfg, widths):
        hb = h[:, c0:c0 + bc]
        if cfg == '1x1':
            outs.append(hb[:, :, ::stride])
        elif cfg[0] == 'max':
            outs.append(F.max_pool2d(hb, (cfg[1], 1), (stride, 1), (1, 0)))
        else:
            k, d = cfg
            pad = (k + (k - 1) * (d - 1) - 1) // 2
            outs.append(F.conv2d(hb, conv_w[ci], conv_b[ci], stride=(stride, 1), padding=(pad, 0), dilation=(d, 1)))
            ci += 1
        c0 += bc
    return torch.cat(outs, 1)


def temporal_branches_bn(z, scale, shift, n_act, branch_cfg, widths, conv_w, conv_b, stride, gamma=None, beta=None,
                         eps=1e-5, want_bn=False):
    """MSTCN's temporal stage (msg3d_utils.py:84-117) after the fused branch 1x1 conv: BN+ReLU on channels < n_act,
    per-branch (k,1) dilated conv / max-pool / strided copy -> o (n,C,T',V), + the train-mode BN of o."""
    h = z * _bc(scale) + _bc(shift)
    h = torch.cat([F.relu(h[:, :n_act]), h[:, n_act:]], 1)
    return _bn_of(_branches(h, branch_cfg, widths, conv_w, conv_b, stride), gamma, beta, eps, want_bn)


def tconv_bn(x1, a1, x2, a2, relu, weight, bias, gamma=None, beta=None, eps=1e-5, want_bn=False, stride=1):
    """Dense (k,1) temporal conv (dilation 1) of the virtual input (see dsgcn_amd.kernels.tconv_bn)."""
    return tconv(virt(x1, a1, x2, a2, relu), weight, bias, stride, 1, gamma, beta, eps, want_bn)


def tconv(h, weight, bias, stride, dilation, gamma=None, beta=None, eps=1e-5, want_bn=False):
    """Dense temporal conv (k,1) of a materialised tensor (unit_tcn, tcn.py:21-27) + the train-mode BN of its output."""
    k = weight.shape[2]
    pad = (k + (k - 1) * (dilation - 1) - 1) // 2
    z = F.conv2d(h, weight, bias, stride=(stride, 1), padding=(pad, 0), dilation=(dilation, 1))
    return _bn_of(z, gamma, beta, eps, want_bn)


def fuse_out(x1, a1, x2, a2, relu, want_tmean=False, tee=False):
    """relu: bool or int flags (bit 0 outer ReLU, bit 1 ReLU on the first term before the add)."""
    relu = int(relu)
    if relu & 2:
        out = virt(x1, a1, None, None, True)
        if x2 is not None:
            out = out + (x2 if a2 is None else x2 * _bc(a2[0]) + _bc(a2[1]))
        if relu & 1:
            out = F.relu(out)
    else:
        out = virt(x1, a1, x2, a2, bool(relu & 1))
    xbar = out.mean(2) if want_tmean else None
    if xbar is not None and want_tmean is not True and int(want_tmean) > xbar.shape[-1]:
        xbar = F.pad(xbar, (0, int(want_tmean) - xbar.shape[-1]))         # padded joint row (kernels.fuse_out)
    return ((out, out, out) if tee else out), xbar


def fuse_out_pool(x1, a1, x2, a2, relu):
    return fuse_out(x1, a1, x2, a2, relu)[0].mean((2, 3))


def temporal_mlp_bn(z, scale, shift, n_act, branch_cfg, widths, conv_w, conv_b, dw_w, dw_b, dw_dil, pw_w, pw_b,
                    merge_after, stride, gamma=None, beta=None, eps=1e-5, want_bn=False):
    """msmlp's temporal stage (see dsgcn_amd.kernels.temporal_mlp_bn) in plain torch ops."""
    n, C, T, V = z.shape
    h = z * _bc(scale) + _bc(shift)
    h = torch.cat([F.relu(h[:, :n_act]), h[:, n_act:]], 1)
    o = _branches(h, branch_cfg, widths, conv_w, conv_b, stride)
    KM = dw_w.shape[1]
    Tout = o.shape[2]
    dw = torch.zeros_like(o)
    for c in range(C):
        dl = int(dw_dil[c])
        if dl == 0:
            continue
        xp = F.pad(h[:, c], (0, 0, (KM - 1) * dl, 0))                     # left pad on the frame axis
        acc = dw_b[c]
        for j in range(KM):
            acc = acc + dw_w[c, j] * xp[:, j * dl:j * dl + (Tout - 1) * stride + 1:stride]
        dw[:, c] = acc
    mix = lambda t: torch.einsum('oc,nctv->notv', pw_w, t) + _bc(pw_b)    # noqa: E731
    out = mix(dw) + o if merge_after else mix(dw + o)
    return _bn_of(out, gamma, beta, eps, want_bn)


def temporal_unitmlp_bn(h, dw_w, dw_b, dw_dil, tw, tb, tdil, pw_w, pw_b, merge_after, stride, gamma=None, beta=None,
                        eps=1e-5, want_bn=False):
    """unitmlp as a whole temporal unit (see dsgcn_amd.kernels.temporal_unitmlp_bn) in plain torch ops."""
    n, C, T, V = h.shape
    KM = dw_w.shape[1]
    Tout = (T + stride - 1) // stride
    dw = h.new_zeros(n, C, Tout, V)
    for c in range(C):
        dl = int(dw_dil[c])
        if dl == 0:
            continue
        xp = F.pad(h[:, c], (0, 0, (KM - 1) * dl, 0))
        acc = dw_b[c]
        for j in range(KM):
            acc = acc + dw_w[c, j] * xp[:, j * dl:j * dl + (Tout - 1) * stride + 1:stride]
        dw[:, c] = acc
    mix = lambda t: torch.einsum('oc,nctv->notv', pw_w.reshape(pw_w.shape[0], -1), t) + _bc(pw_b)    # noqa: E731
    t = tconv(h, tw, tb, stride, tdil)[0] if tw is not None else None
    if t is None:
        out = mix(dw)
    else:
        out = mix(dw) + t if merge_after else mix(dw + t)
    return _bn_of(out, gamma, beta, eps, want_bn)


def bn_running_update(items):
    """kernels.bn_running_update in plain torch ops."""
    with torch.no_grad():
        for bn, mean, var, c in items:
            m = float(bn.momentum)
            bn.num_batches_tracked += 1
            bn.running_mean.mul_(1.0 - m).add_(mean, alpha=m)
            bn.running_var.mul_(1.0 - m).add_(var * (c / max(c - 1.0, 1.0)), alpha=m)


def head_loss(feat, weight, bias, label, persons, loss_weight=1.0):
    """kernels.head_loss in plain torch ops: person mean, Linear, cross entropy, top-1 / top-5 accuracy (the rank of the
    label under a stable ascending argsort)."""
    N = feat.shape[0] // persons
    score = F.linear(feat.reshape(N, persons, -1).mean(1), weight, bias)
    loss = F.cross_entropy(score, label) * loss_weight
    with torch.no_grad():
        sl = score.gather(1, label.view(-1, 1))
        idx = torch.arange(score.shape[1], device=score.device)[None]
        rank = ((score > sl) | ((score == sl) & (idx > label.view(-1, 1)))).sum(1)
        acc = torch.stack([(rank < 1).double().mean(), (rank < 5).double().mean()])
    return loss, acc, score.detach()


def data_bn_eligible(x, bn):
    return isinstance(bn, torch.nn.BatchNorm1d) and x.dim() == 5


def data_bn(x, bn, bn_type):
    """kernels.data_bn as the reference writes it (dgstgcn.py:158-164)."""
    N, M, T, V, C = x.shape
    x = x.permute(0, 1, 3, 4, 2).contiguous()
    x = bn(x.view(N, M * V * C, T)) if bn_type == 'MVC' else bn(x.view(N * M, V * C, T))
    return x.view(N, M, V, C, T).permute(0, 1, 3, 4, 2).contiguous().view(N * M, C, T, V)
