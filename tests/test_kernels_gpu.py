"""-m gpu: every HIP op through the C ABI vs the plain-PyTorch statement of the same op
(tests/torch_ops.py) evaluated in fp64 on the same inputs.  Tolerances are written per test."""
import pytest
import torch

import dsgcn_amd
from dsgcn_amd import kernels as K
import torch_ops as R

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def maxabs(a, b):
    return (a.double() - b.double()).abs().max().item()


@pytest.mark.parametrize('n,KC,T,V,relu,affine', [
    (3, 24, 64, 25, True, True), (2, 48, 32, 25, True, True), (2, 96, 16, 25, True, True),
    (2, 24, 100, 17, True, True), (2, 10, 25, 17, True, True), (2, 6, 64, 25, False, False),
    (1, 5, 7, 25, True, True), (2, 4, 130, 18, True, True)])
def test_aggregate(n, KC, T, V, relu, affine):
    g = torch.Generator().manual_seed(n * 1000 + KC + T)
    zp = torch.randn(n, KC, T, V, generator=g)
    ahat = torch.randn(n, KC, V, V, generator=g) * 0.3
    sc = torch.randn(KC, generator=g) if affine else None
    sh = torch.randn(KC, generator=g) * 0.5 if affine else None
    dy = torch.randn(n, KC, T, V, generator=g)

    def run(mod, dt, dev):
        t = [x.to(dev, dt).requires_grad_() if x is not None else None for x in (zp, ahat, sc, sh)]
        y = mod.aggregate(t[0], (t[2], t[3]) if affine else None, relu, t[1])
        y.backward(dy.to(dev, dt))
        return [y] + [x.grad if x is not None else None for x in t]

    got = run(K, torch.float32, DEV)
    ref = run(R, torch.float64, 'cpu')
    names = ['y', 'dzp', 'dahat', 'dscale', 'dshift']
    for nm, a, b in zip(names, got, ref):
        if b is None:
            continue
        # fp32 accumulation over <= V (fwd) / T (dahat) / n*T*V (dscale) terms: 2e-6 relative L2
        assert rel(a.cpu(), b) < 2e-6, (nm, rel(a.cpu(), b))


def _dyn_inputs(n, Ci, mid, V, layout, seed=0):
    g = torch.Generator().manual_seed(seed)
    gr = dsgcn_amd.Graph(layout=layout, mode='spatial')
    P, E = 5, 15
    nt = torch.tensor(gr.node_type, dtype=torch.int32)
    et = torch.tensor(gr.edge_type, dtype=torch.int32)
    t = dict(
        xbar=torch.randn(n, Ci, V, generator=g),
        A=torch.randn(3, V, V, generator=g) * 0.02 + 0.04,
        alpha=torch.randn(3, generator=g) * 0.5, beta=torch.randn(3, generator=g) * 0.5,
        w1=torch.randn(2 * mid, Ci, generator=g) / Ci ** 0.5, b1=torch.randn(2 * mid, generator=g) * 0.1,
        w2=torch.randn(2 * mid, Ci, generator=g) / Ci ** 0.5, b2=torch.randn(2 * mid, generator=g) * 0.1,
        wse=torch.randn(mid * P, Ci, generator=g) / Ci ** 0.5, bse=torch.randn(mid * P, generator=g) * 0.1,
        we=torch.randn(E * mid, mid, generator=g) / mid ** 0.5, be=torch.randn(E * mid, generator=g) * 0.1)
    return t, nt, et


@pytest.mark.parametrize('n,Ci,mid,V,layout', [
    (3, 3, 8, 25, 'nturgb+d'), (2, 64, 8, 25, 'nturgb+d'), (2, 64, 16, 25, 'nturgb+d'),
    (2, 128, 32, 25, 'nturgb+d'), (2, 256, 32, 25, 'nturgb+d'), (2, 64, 8, 17, 'coco')])
def test_dynadj(n, Ci, mid, V, layout):
    t, nt, et = _dyn_inputs(n, Ci, mid, V, layout, seed=Ci + mid)
    g = torch.Generator().manual_seed(7)
    dah = torch.randn(n, 3 * mid, V, V, generator=g)
    order = list(t)

    def run(mod, dt, dev):
        tt = {k: v.to(dev, dt).requires_grad_() for k, v in t.items()}
        out = mod.dynadj(*[tt[k] for k in order], nt.to(dev), et.to(dev))
        out.backward(dah.to(dev, dt))
        return out, {k: v.grad for k, v in tt.items()}

    out, grads = run(K, torch.float32, DEV)
    ro, rg = run(R, torch.float64, 'cpu')
    # forward: tanh/exp in fp32 (ocml, ~1-2 ulp) + <=256-term dot products
    assert rel(out.cpu(), ro) < 2e-6, rel(out.cpu(), ro)
    for k in order:
        # fp32 chain rule with float atomics on the weight grads: 2e-5 relative L2
        assert rel(grads[k].cpu(), rg[k]) < 2e-5, (k, rel(grads[k].cpu(), rg[k]))
