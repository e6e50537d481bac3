"""Where does the dense temporal conv (csrc/tcg.hip) differ from fp64 at full size?  python tools/tcg_diag.py n Ci Co T V KT mode stride"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from dsgcn_amd import kernels as K
import torch_ops as R

n, Ci, Co, T, V, KT = [int(a) for a in sys.argv[1:7]]
mode, stride = sys.argv[7], int(sys.argv[8])
g = torch.Generator().manual_seed(Ci * 3 + Co + T + KT)
rnd = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale
x1 = rnd(n, Ci, T, V)
a1 = a2 = x2 = None
relu = False
if mode in ('res_plain', 'res_affine', 'affine_relu'):
    a1 = (torch.rand(Ci, generator=g) + 0.5, rnd(Ci, scale=0.3)); relu = True
if mode in ('res_plain', 'res_affine'):
    x2 = rnd(n, Ci, T, V)
if mode == 'res_affine':
    a2 = (torch.rand(Ci, generator=g) + 0.5, rnd(Ci, scale=0.3))
w = rnd(Co, Ci, KT, 1, scale=(Ci * KT) ** -0.5); b = rnd(Co, scale=0.1)
gamma = torch.rand(Co, generator=g) + 0.5; beta = rnd(Co, scale=0.2)
gz = rnd(n, Co, (T + stride - 1) // stride, V); gsc, gsh = rnd(Co), rnd(Co)

def run(mod, dt):
    mk = lambda t: None if t is None else t.to('cuda', dt).requires_grad_()
    tx1, tx2, tw, tb, tg, tbeta = mk(x1), mk(x2), mk(w), mk(b), mk(gamma), mk(beta)
    ta1 = None if a1 is None else (mk(a1[0]), mk(a1[1])); ta2 = None if a2 is None else (mk(a2[0]), mk(a2[1]))
    z, sc, sh, mean, var = mod.tconv_bn(tx1, ta1, tx2, ta2, relu, tw, tb, tg, tbeta, 1e-5, True, stride)
    ((z * gz.to('cuda', dt)).sum() + (sc * gsc.to('cuda', dt)).sum() + (sh * gsh.to('cuda', dt)).sum()).backward()
    o = dict(z=z, dx1=tx1.grad, dw=tw.grad)
    if tx2 is not None: o['dx2'] = tx2.grad
    return {k: v.detach() for k, v in o.items()}

got, ref = run(K, torch.float32), run(R, torch.float64)
for k in got:
    d = (got[k].double() - ref[k]).abs()
    rel = (got[k].double() - ref[k]).norm() / ref[k].norm()
    print(k, 'rel', float(rel), 'max', float(d.max()), 'ref rms', float(ref[k].pow(2).mean().sqrt()))
    if d.dim() == 4 and k != 'dw':
        bad = d > 1e-4 * float(ref[k].abs().max())
        print('  bad elements', int(bad.sum()), 'of', bad.numel())
        if bad.any():
            idx = bad.nonzero()
            for name, col in zip('nctv', range(4)):
                u = torch.unique(idx[:, col])
                print('   ', name, 'count', len(u), u[:40].tolist())
