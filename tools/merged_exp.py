"""What one 64-row block per position tile would cost: the C=64 / C=128 layer shapes with their four conv windows merged
into ONE window of the same total width (timing only; a different op)."""
import os, sys
sys.path.insert(0, '/root/repo')
import torch
from dsgcn_amd import kernels as K, native
n = 128
dev = torch.device('cuda')
for C, T, wconv in ((64, 64, 44), (128, 32, 64)):
    rest = C - wconv
    cfg = [(3, 1), ('max', 3), '1x1']
    widths = [wconv, rest // 2, rest - rest // 2]
    n_act = C - widths[2]
    g = torch.Generator().manual_seed(0)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
    z, zaug = r(n, C, T, 25).requires_grad_(), r(n, C, T).requires_grad_()
    scale = torch.cat([torch.rand(n_act, generator=g) + 0.5, torch.ones(widths[2])]).to(dev).requires_grad_()
    shift = torch.cat([torch.randn(n_act, generator=g) * 0.3, torch.zeros(widths[2])]).to(dev).requires_grad_()
    cw = [r(wconv, wconv, 3, 1, scale=(3 * wconv) ** -0.5).requires_grad_()]
    cb = [r(wconv, scale=0.1).requires_grad_()]
    coeff = r(25, scale=0.5).requires_grad_()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev).requires_grad_(), r(C, scale=0.2).requires_grad_()
    gf = r(n, C, T, 25)
    for _ in range(12):
        out = K.temporal_ms(z, zaug, scale, shift, n_act, cfg, widths, cw, cb, coeff, 1, gamma, beta, 1e-5, True)
        ((out[0] * gf).sum() + out[1].sum() + out[2].sum()).backward()
    torch.cuda.synchronize()
    print('done', C, T, widths)
