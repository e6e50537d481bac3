"""K-C micro-benchmark: forward / data gradient / weight gradient per DS-STGCN layer shape, HIP-event timed through the
C ABI (no autograd, no Python between launches beyond the ctypes call).  Usage: kc_bench.py [variant ...] where a variant
is key=value pairs joined by commas for dsgcn_pwconv_tuning (e.g. 3=0  or  3=3,4=2,6=16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsgcn_amd import native
lib = native.lab_lib(); dev = 'cuda'; st = torch.cuda.current_stream().cuda_stream
n, V = 128, int(os.environ.get('KC_V', '25'))
SHAPES = [('pre1', 64, 24, 64, 0), ('post1', 24, 64, 64, 1), ('branch1', 64, 64, 64, 2), ('transf1', 64, 64, 64, 1),
          ('pre5', 128, 48, 32, 0), ('post5', 48, 128, 32, 1), ('branch5', 128, 128, 32, 2), ('transf5', 128, 128, 32, 1),
          ('pre8', 256, 96, 16, 0), ('post8', 96, 256, 16, 1), ('branch8', 256, 256, 16, 2), ('transf8', 256, 256, 16, 1)]


if os.environ.get('KC_SHAPES'):          # name,Ci,Co,T,mode;...
    SHAPES = [(a, int(b), int(c), int(d), int(e)) for a, b, c, d, e in (x.split(',') for x in os.environ['KC_SHAPES'].split(';'))]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def run(variant):
    keys = {3: 15, 4: 0, 5: 0, 6: 0, 7: 512, 8: 32, 9: 512, 14: 1}
    if variant:
        for kv in variant.split(','):
            k, v = kv.split('=')
            keys[int(k)] = int(v)
    for k, v in keys.items():
        assert lib.dsgcn_pwconv_tuning(k, v) == 0
    print(f'--- variant {variant or "default"}')
    print(f'{"name":8s} {"Ci":>4s} {"Co":>4s} | {"fwd us":>7s} {"TB/s":>5s} {"TF":>5s} | {"dgrad":>7s} {"TB/s":>5s} {"TF":>5s} | {"wgrad":>7s} {"TB/s":>5s} {"TF":>5s}')
    tot = [0.0, 0.0, 0.0]
    for name, Ci, Co, T, mode in SHAPES:
        x1 = torch.randn(n, Ci, T, V, device=dev)
        x2 = torch.randn(n, Ci, T, V, device=dev) if mode == 2 else None
        s1 = (torch.rand(Ci, device=dev) + .5) if mode else None
        h1 = (torch.randn(Ci, device=dev) * .1) if mode else None
        relu = 1 if mode else 0
        w = torch.randn(Co, Ci, device=dev) * Ci ** -.5; b = torch.zeros(Co, device=dev)
        z = torch.empty(n, Co, T, V, device=dev); gz = torch.randn(n, Co, T, V, device=dev)
        A0 = torch.randn(Co, device=dev) * 1e-3; B0 = torch.randn(Co, device=dev) * 1e-3
        dx = torch.empty_like(x1); dx2 = torch.empty_like(x1) if mode == 2 else None
        part = torch.empty(lib.dsgcn_pwconv_partial_rows(n, Ci, Co, T, V, 1, 0), Co, 2, device=dev)
        ipart = torch.empty(lib.dsgcn_pwconv_ipart_rows(n, Ci, Co, T, V, 1), Ci, 3, device=dev) if mode else None
        splits = lib.dsgcn_pwconv_wgrad_splits(n, Ci, Co, T, V, 1)
        pstride = Co * Ci + Co
        wpart = torch.empty(splits, pstride, device=dev)
        P = lambda t: None if t is None else t.data_ptr()
        wsb = lib.dsgcn_pwconv_wsplit_bytes(n, Ci, Co, T, V, 1)      # (14=0: no pre-split weight image)
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev) if wsb else None
        if ws is not None:
            assert lib.dsgcn_pwconv_wsplit(P(w), Ci, Co, P(ws), st) == 0

        def fwd():
            assert lib.dsgcn_pwconv_fwd_ws(P(x1), P(s1), P(h1), P(x2), None, None, relu, P(w), P(b), P(z), None, P(part),
                                           n, Ci, Co, T, V, 1, 0, 1, P(ws), st) == 0

        def dgrad():
            assert lib.dsgcn_pwconv_dgrad_ws(P(x1), P(s1), P(h1), P(x2), None, None, relu, P(w), P(z), None, P(gz), None,
                                             P(A0), P(B0), P(dx), P(dx2), P(ipart), n, Ci, Co, T, V, 1, 0, P(ws), st) == 0

        def wgrad():
            assert lib.dsgcn_pwconv_wgrad(P(x1), P(s1), P(h1), P(x2), None, None, relu, P(z), None, P(gz), None, P(A0),
                                          P(B0), wpart.data_ptr(), wpart.data_ptr() + 4 * Co * Ci, pstride, n, Ci, Co,
                                          T, V, 1, 0, st) == 0
        rows_f = lib.dsgcn_pwconv_bwd_rows(n, Ci, Co, T, V, 1)
        if rows_f:
            wpf = torch.empty(rows_f, pstride, device=dev)
            ipf = torch.empty(rows_f, Ci, 3, device=dev) if mode else None

            def fused():
                assert lib.dsgcn_pwconv_bwd(P(x1), P(s1), P(h1), P(x2), None, None, relu, P(w), P(z), P(gz), P(A0), P(B0), P(dx),
                                            P(dx2), P(ipf), wpf.data_ptr(), wpf.data_ptr() + 4 * Co * Ci, pstride, n, Ci, Co, T, V, st) == 0
        L = n * T * V
        nin = 2 if mode == 2 else 1
        flops = 2.0 * Ci * Co * L
        res = []
        for i, (fn, byts) in enumerate(((fwd, 4 * L * (Ci * nin + Co)),
                                        (dgrad, 4 * L * (2 * Co + Ci * nin * (2 if mode else 1))),
                                        (wgrad, 4 * L * (2 * Co + Ci * nin)))):
            t = timeit(fn)
            tot[i] += t
            res.append(f'{t:7.1f} {byts / t / 1e6:5.2f} {flops / t / 1e6:5.1f}')
        if rows_f:
            t = timeit(fused)
            res.append(f'fused bwd {t:6.1f} us {4 * L * (2 * Co + 2 * Ci) / t / 1e6:5.2f} TB/s')
            if os.environ.get('KC_PHASES'):
                import numpy as np
                torch.cuda.synchronize(); fused(); torch.cuda.synchronize()
                ph = np.zeros(64, dtype=np.int64)
                assert lib.dsgcn_bwd64_phases(ph.ctypes.data) == 0
                k_ = int(ph[63]); d = np.diff(ph[:k_]) / 100.0
                un = d[1:-2].reshape(-1, 2)
                res.append('| wg0: tables %.1f; per unit (previous finish + commit)/products: ' % d[0] + ' '.join(f'{a_:.2f}/{b_:.2f}' for a_, b_ in un) + f'; last finish {d[-2]:.2f}; rows {d[-1]:.1f}; total {(ph[k_ - 1] - ph[0]) / 100.0:.1f} us')
        print(f'{name:8s} {Ci:4d} {Co:4d} | ' + ' | '.join(res), flush=True)
        if os.environ.get('KC_PHASES') and ws is not None and keys[14]:
            import numpy as np
            for nm, fn in (('fwd', fwd), ('dgrad', dgrad)):
                torch.cuda.synchronize(); fn(); torch.cuda.synchronize()
                ph = np.zeros(64, dtype=np.int64)
                assert lib.dsgcn_pwg2_phases(ph.ctypes.data) == 0
                k_ = int(ph[63]); d = np.diff(ph[:k_]) / 100.0
                nchunk = (k_ - 4) // 3
                ch = d[1:1 + 3 * nchunk].reshape(nchunk, 3)
                print(f'    {nm}: prologue {d[0]:.1f} us | per chunk commit/barrier/products (us): ' +
                      ' '.join(f'{a_:.2f}/{b_:.2f}/{c_:.2f}' for a_, b_, c_ in ch) + f' | drain {d[-2]:.1f} epilogue {d[-1]:.1f} | total {(ph[k_ - 1] - ph[0]) / 100.0:.1f}')
    print(f'total us: fwd {tot[0]:.0f} dgrad {tot[1]:.0f} wgrad {tot[2]:.0f}')


if __name__ == '__main__':
    for v in (sys.argv[1:] or ['']):
        run(v)
