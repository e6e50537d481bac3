"""K-C accuracy check: the three-term bf16 form (B3) and the fp32 MFMA form of the 1x1 conv forward / data gradient /
weight gradient against an fp64 evaluation of the same inputs (relative L2 error of each: z, column sums, dx, dW, db,
[dx2], [input sums]), through the lab library's tuning keys 10 (GEMM form: forward / data gradient bits) and 13 (weight gradient)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsgcn_amd import native
lib = native.lab_lib(); dev = 'cuda'; st = torch.cuda.current_stream().cuda_stream
n, V = int(os.environ.get('KC_N', 16)), int(os.environ.get('KC_V', 25))
SHAPES = [('pre5', 128, 48, 32, 0), ('post5', 48, 128, 32, 1), ('branch5', 128, 128, 32, 2), ('transf5', 128, 128, 32, 1),
          ('pre8', 256, 96, 16, 0), ('post8', 96, 256, 16, 1), ('branch8', 256, 256, 16, 2), ('transf8', 256, 256, 16, 1),
          ('odd', 112, 200, 7, 2), ('odd2', 70, 130, 12, 1), ('down5', 64, 128, 32, 0), ('proj', 256, 96, 1, 0)]
P = lambda t: None if t is None else t.data_ptr()


def rel(a, b):
    return ((a.double() - b).norm() / b.norm()).item()


torch.manual_seed(0)
for name, Ci, Co, T, mode in SHAPES:
    x1 = torch.randn(n, Ci, T, V, device=dev)
    x2 = torch.randn(n, Ci, T, V, device=dev) if mode == 2 else None
    s1 = (torch.rand(Ci, device=dev) + .5) if mode else None
    h1 = (torch.randn(Ci, device=dev) * .1) if mode else None
    relu = 1 if mode else 0
    w = torch.randn(Co, Ci, device=dev) * Ci ** -.5; b = torch.randn(Co, device=dev)
    gz = torch.randn(n, Co, T, V, device=dev)
    A0 = torch.randn(Co, device=dev) * 1e-1; B0 = torch.randn(Co, device=dev) * 1e-1
    # fp64 reference
    v = x1.double()
    if mode:
        v = v * s1.double().view(1, -1, 1, 1) + h1.double().view(1, -1, 1, 1)
        if mode == 2:
            v = v + x2.double()
        pre = v
        v = v.clamp_min(0)
    zr = torch.einsum('oc,nctv->notv', w.double(), v) + b.double().view(1, -1, 1, 1)
    out = {}
    for b3 in (0, 1):
        assert lib.dsgcn_pwconv_tuning(12, int(os.environ.get('KC_MINL', 128))) == 0 and lib.dsgcn_pwconv_tuning(10, 3 * b3) == 0 and lib.dsgcn_pwconv_tuning(13, b3) == 0
        z = torch.empty(n, Co, T, V, device=dev)
        part = torch.empty(lib.dsgcn_pwconv_partial_rows(n, Ci, Co, T, V, 1, 0), Co, 2, device=dev)
        assert lib.dsgcn_pwconv_fwd(P(x1), P(s1), P(h1), P(x2), None, None, relu, P(w), P(b), P(z), None, P(part),
                                    n, Ci, Co, T, V, 1, 0, 1, st) == 0
        dx = torch.empty_like(x1); dx2 = torch.empty_like(x1) if mode == 2 else None
        ipart = torch.empty(lib.dsgcn_pwconv_ipart_rows(n, Ci, Co, T, V, 1), Ci, 3, device=dev) if mode else None
        assert lib.dsgcn_pwconv_dgrad(P(x1), P(s1), P(h1), P(x2), None, None, relu, P(w), P(z), None, P(gz), None,
                                      P(A0), P(B0), P(dx), P(dx2), P(ipart), n, Ci, Co, T, V, 1, 0, st) == 0
        torch.cuda.synchronize()
        dze = gz.double() + A0.double().view(1, -1, 1, 1) + B0.double().view(1, -1, 1, 1) * z.double()
        dv = torch.einsum('oc,notv->nctv', w.double(), dze)
        if mode:
            dv = dv * (pre > 0)
            dxr = dv * s1.double().view(1, -1, 1, 1)
        else:
            dxr = dv
        splits = lib.dsgcn_pwconv_wgrad_splits(n, Ci, Co, T, V, 1)
        pstride = Co * Ci + Co
        wpart = torch.empty(splits, pstride, device=dev)
        assert lib.dsgcn_pwconv_wgrad(P(x1), P(s1), P(h1), P(x2), None, None, relu, P(z), None, P(gz), None, P(A0),
                                      P(B0), wpart.data_ptr(), wpart.data_ptr() + 4 * Co * Ci, pstride, n, Ci, Co,
                                      T, V, 1, 0, st) == 0
        torch.cuda.synchronize()
        dwr = torch.einsum('notv,nctv->oc', dze, v)
        dw = wpart.double().sum(0)
        r = [rel(z, zr), rel(part.double().sum(0)[:, 0], zr.sum((0, 2, 3))), rel(dx, dxr),
             rel(dw[:Co * Ci].view(Co, Ci), dwr), rel(dw[Co * Ci:], dze.sum((0, 2, 3)))]
        if mode == 2:
            r.append(rel(dx2, dv))
        if mode:
            r.append(rel(ipart.double().sum(0)[:, 1], dv.sum((0, 2, 3))))
        out[b3] = r
    print(f'{name:8s} {Ci:4d}->{Co:4d}  fp32 mfma: ' + ' '.join(f'{e:.2e}' for e in out[0]) + '   b3: ' + ' '.join(f'{e:.2e}' for e in out[1]), flush=True)
