"""Fused ops of the DS-GCN hot path, each a ``torch.autograd.Function`` over the C-ABI HIP library.

PyTorch is plumbing here (device memory, the current stream, autograd bookkeeping); every byte of
arithmetic in these ops happens in ``libdsgcn.so`` (``ds-gcn_amd/csrc``).  There is NO CPU or
eager fallback: inputs must be CUDA fp32 tensors and the library must load, otherwise the op raises.

``ops()`` returns the active op namespace (this module).  ``use_ops(ns)`` is a test seam that swaps
in another namespace with the same signatures (``tests/torch_ops.py``) so the host-side wiring can be
checked against the oracle without a GPU; the product never calls it.
"""
import contextlib
import sys

_active = None


def ops():
    return _active if _active is not None else sys.modules[__name__]


@contextlib.contextmanager
def use_ops(namespace):
    global _active
    prev = _active
    _active = namespace
    try:
        yield namespace
    finally:
        _active = prev


NAME = 'hip'

import torch
import torch.nn.functional as F

from . import native


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('dsgcn: the hot path runs only as HIP kernels on CUDA/ROCm tensors; got a CPU tensor '
                               '(there is no CPU fallback)')


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise TypeError(f'dsgcn kernels compute in fp32, got {t.dtype}')
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------
# K-A  gather-aggregate
# ---------------------------------------------------------------------------------------------

class _Aggregate(torch.autograd.Function):

    @staticmethod
    def forward(ctx, zp, scale, shift, relu, ahat):
        _require_cuda(zp, ahat)
        zp, ahat, scale, shift = _f32c(zp), _f32c(ahat), _f32c(scale), _f32c(shift)
        n, KC, T, V = zp.shape
        assert ahat.shape == (n, KC, V, V), (ahat.shape, zp.shape)
        y = torch.empty_like(zp)
        rc = native.lib().dsgcn_aggregate_fwd(_ptr(zp), _ptr(scale), _ptr(shift), int(relu), _ptr(ahat), _ptr(y),
                                              n, KC, T, V, _stream())
        native.check(rc, 'dsgcn_aggregate_fwd')
        ctx.save_for_backward(zp, scale, shift, ahat)
        ctx.relu = int(relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        zp, scale, shift, ahat = ctx.saved_tensors
        n, KC, T, V = zp.shape
        dy = _f32c(dy)
        dzp = torch.empty_like(zp)
        dahat = torch.empty_like(ahat)
        partial = torch.empty((n, KC, 2), device=zp.device, dtype=torch.float32)
        rc = native.lib().dsgcn_aggregate_bwd(_ptr(zp), _ptr(scale), _ptr(shift), ctx.relu, _ptr(ahat), _ptr(dy),
                                              _ptr(dzp), _ptr(dahat), _ptr(partial), n, KC, T, V, _stream())
        native.check(rc, 'dsgcn_aggregate_bwd')
        dscale = dshift = None
        if scale is not None:
            red = partial.sum(0)
            dscale, dshift = red[:, 0], red[:, 1]
        return dzp, dscale, dshift, None, dahat


def aggregate(zp, ap, relu, ahat):
    scale, shift = ap if ap is not None else (None, None)
    return _Aggregate.apply(zp, scale, shift, relu, ahat)


# ---------------------------------------------------------------------------------------------
# K-B  dynamic adjacency
# ---------------------------------------------------------------------------------------------

_pair_cache = {}


def _edge_class_lists(edge_type):
    """Joint pairs sorted by edge class + class offsets (device int32), cached per edge_type tensor."""
    key = (edge_type.data_ptr(), edge_type.device)
    hit = _pair_cache.get(key)
    if hit is None:
        et = edge_type.flatten().to(torch.int64)
        order = torch.argsort(et, stable=True).to(torch.int32)
        E = int(et.max().item()) + 1
        counts = torch.bincount(et, minlength=E)
        start = torch.zeros(E + 1, dtype=torch.int32, device=edge_type.device)
        start[1:] = torch.cumsum(counts, 0).to(torch.int32)
        hit = (order.contiguous(), start.contiguous(), E)
        _pair_cache[key] = hit
    return hit


class _DynAdj(torch.autograd.Function):

    @staticmethod
    def forward(ctx, xbar, A, alpha, beta, w1, b1, w2, b2, wse, bse, we, be, node_type, edge_type):
        _require_cuda(xbar, A)
        xbar, A, alpha, beta, w1, b1, w2, b2, wse, bse, we, be = [
            _f32c(t) for t in (xbar, A, alpha, beta, w1, b1, w2, b2, wse, bse, we, be)]
        n, Ci, V = xbar.shape
        mid = w1.shape[0] // 2
        P = wse.shape[0] // mid
        E = we.shape[0] // mid
        assert A.shape[0] == 3 and node_type.dtype == torch.int32 and edge_type.dtype == torch.int32
        ahat = torch.empty((n, 3 * mid, V, V), device=xbar.device, dtype=torch.float32)
        rc = native.lib().dsgcn_dynadj_fwd(
            _ptr(xbar), _ptr(A), _ptr(alpha), _ptr(beta), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(wse),
            _ptr(bse), _ptr(we), _ptr(be), _ptr(node_type), _ptr(edge_type), _ptr(ahat), n, Ci, mid, V, P, E,
            _stream())
        native.check(rc, 'dsgcn_dynadj_fwd')
        ctx.save_for_backward(xbar, A, alpha, beta, w1, b1, w2, b2, wse, bse, we, be, node_type, edge_type)
        ctx.dims = (n, Ci, mid, V, P, E)
        return ahat

    @staticmethod
    def backward(ctx, dahat):
        xbar, A, alpha, beta, w1, b1, w2, b2, wse, bse, we, be, node_type, edge_type = ctx.saved_tensors
        n, Ci, mid, V, P, E = ctx.dims
        dahat = _f32c(dahat)
        dev = xbar.device
        order, start, _ = _edge_class_lists(edge_type)
        dd = torch.empty_like(dahat)
        dproj = torch.empty((n, 5 * mid, V), device=dev, dtype=torch.float32)
        dxbar = torch.empty_like(xbar)
        pA = torch.empty((n, 3, V, V), device=dev, dtype=torch.float32)
        pab = torch.empty((n, 6), device=dev, dtype=torch.float32)
        zeros = torch.zeros(E * mid * mid + E * mid + 9 * mid * Ci + 9 * mid, device=dev, dtype=torch.float32)
        dwe, dbe, dwp, dbp = torch.split(zeros, [E * mid * mid, E * mid, 9 * mid * Ci, 9 * mid])
        rc = native.lib().dsgcn_dynadj_bwd(
            _ptr(xbar), _ptr(alpha), _ptr(beta), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(wse), _ptr(bse),
            _ptr(we), _ptr(be), _ptr(node_type), _ptr(edge_type), _ptr(order), _ptr(start), _ptr(dahat), _ptr(dd),
            _ptr(dproj), _ptr(dxbar), _ptr(pA), _ptr(pab), _ptr(dwe), _ptr(dbe), _ptr(dwp), _ptr(dbp),
            n, Ci, mid, V, P, E, _stream())
        native.check(rc, 'dsgcn_dynadj_bwd')
        dA = pA.sum(0)
        dab = pab.sum(0)
        dwp = dwp.view(9 * mid, Ci)
        return (dxbar, dA, dab[:3], dab[3:], dwp[:2 * mid], dbp[:2 * mid], dwp[2 * mid:4 * mid], dbp[2 * mid:4 * mid],
                dwp[4 * mid:], dbp[4 * mid:], dwe.view(E * mid, mid), dbe, None, None)


def dynadj(xbar, A, alpha, beta, w1, b1, w2, b2, wse, bse, we, be, node_type, edge_type):
    return _DynAdj.apply(xbar, A, alpha, beta, w1, b1, w2, b2, wse, bse, we, be, node_type, edge_type)


# ---------------------------------------------------------------------------------------------
# INTERIM: ops below still run as PyTorch-ROCm device ops (MIOpen/ATen on the GPU) until their HIP
# kernels (K-C pwconv, K-D temporal_ms, fuse_out) land; they are CUDA-only like everything else here.
# ---------------------------------------------------------------------------------------------

def _bc(p):
    return p[None, :, None, None]


def _virt(x1, a1, x2, a2, relu):
    v = x1 if a1 is None else x1 * _bc(a1[0]) + _bc(a1[1])
    if x2 is not None:
        v = v + (x2 if a2 is None else x2 * _bc(a2[0]) + _bc(a2[1]))
    return F.relu(v) if relu else v


def pwconv(x1, a1, x2, a2, relu, weight, bias, stride=1, aug=False, stats=True):
    _require_cuda(x1)
    v = _virt(x1, a1, x2, a2, relu)
    if stride != 1:
        v = v[:, :, ::stride]
    z = F.conv2d(v, weight.reshape(weight.shape[0], -1, 1, 1), bias)
    zaug = z.mean(-1) if aug else None
    mean = var = None
    if stats:
        full = torch.cat([z, zaug[..., None]], -1) if aug else z
        var, mean = torch.var_mean(full, (0, 2, 3), unbiased=False)
    return z, zaug, mean, var


def bn_affine(mean, var, weight, bias, eps):
    scale = weight * torch.rsqrt(var + eps)
    return scale, bias - mean * scale


def tmean(x):
    _require_cuda(x)
    return x.mean(2)


def temporal_ms(z, zaug, scale, shift, n_act, branch_cfg, widths, conv_w, conv_b, add_coeff, stride, stats=True):
    _require_cuda(z)
    n, C, T, V = z.shape
    full = torch.cat([z, zaug[..., None]], -1)
    h = full * _bc(scale) + _bc(shift)
    h = torch.cat([F.relu(h[:, :n_act]), h[:, n_act:]], 1)
    outs, c0, ci = [], 0, 0
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
    mean = var = None
    if stats:
        var, mean = torch.var_mean(f, (0, 2, 3), unbiased=False)
    return f, mean, var


def tconv(x1, a1, relu, weight, bias, stride, dilation, stats=True):
    _require_cuda(x1)
    v = _virt(x1, a1, None, None, relu)
    k = weight.shape[2]
    pad = (k + (k - 1) * (dilation - 1) - 1) // 2
    z = F.conv2d(v, weight, bias, stride=(stride, 1), padding=(pad, 0), dilation=(dilation, 1))
    mean = var = None
    if stats:
        var, mean = torch.var_mean(z, (0, 2, 3), unbiased=False)
    return z, mean, var


def aggregate_shared(zp, A, K, stats=True):
    _require_cuda(zp)
    n, KC, T, V = zp.shape
    y = torch.einsum('nkctv,kvw->nctw', zp.view(n, K, KC // K, T, V), A)
    mean = var = None
    if stats:
        var, mean = torch.var_mean(y, (0, 2, 3), unbiased=False)
    return y, mean, var


def fuse_out(x1, a1, x2, a2, relu, want_tmean=False):
    _require_cuda(x1)
    out = _virt(x1, a1, x2, a2, relu)
    return out, (out.mean(2) if want_tmean else None)
