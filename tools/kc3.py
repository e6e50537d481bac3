"""K-C wide convs, pre-split forms side by side: accuracy against fp64 and HBM-cold launch times of the forward / data
gradient through dsgcn_pwconv_fwd_ws / _dgrad_ws, for each value of lab key 14 given on the command line (1 = k_pwg2,
row-major weight image through LDS; 2 = k_pwg3, fragment-order image straight into registers, two workgroups per CU).
Usage: kc3.py [1 2 ...]   env: KC_N (samples, default 128), KC_CHECK_N (samples of the accuracy pass, default 6)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsgcn_amd import native
lib = native.lab_lib(); dev = 'cuda'; st = torch.cuda.current_stream().cuda_stream
V = int(os.environ.get('KC_V', 25))
# name, Ci, Co, T, mode (0 plain, 1 affine+relu, 2 two streams)
SHAPES = [('pre5', 128, 48, 32, 0), ('post5', 48, 128, 32, 1), ('branch5', 128, 128, 32, 2), ('transf5', 128, 128, 32, 1),
          ('pre8', 256, 96, 16, 0), ('post8', 96, 256, 16, 1), ('branch8', 256, 256, 16, 2), ('transf8', 256, 256, 16, 1),
          ('down5', 64, 128, 32, 0), ('odd', 112, 200, 7, 2), ('odd2', 72, 132, 12, 1), ('wide', 320, 300, 8, 1)]
if os.environ.get('KC_SHAPES'):
    SHAPES = [(a, int(b), int(c), int(d), int(e)) for a, b, c, d, e in (x.split(',') for x in os.environ['KC_SHAPES'].split(';'))]
P = lambda t: None if t is None else t.data_ptr()
_seen_stamps = set()


def rel(a, b):
    return ((a.double() - b).norm() / b.norm()).item()


def operands(n, Ci, Co, T, mode, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    d = dict(x1=r(n, Ci, T, V), x2=r(n, Ci, T, V) if mode == 2 else None, gz=r(n, Co, T, V),
             z=torch.empty(n, Co, T, V, device=dev), dx=torch.empty(n, Ci, T, V, device=dev),
             dx2=torch.empty(n, Ci, T, V, device=dev) if mode == 2 else None)
    return d


def run_form(key14, n, Ci, Co, T, mode, par, q):
    s1, h1, w, b, A0, B0 = par
    relu = 1 if mode else 0
    assert lib.dsgcn_pwconv_tuning(14, key14) == 0
    wsb = lib.dsgcn_pwconv_wsplit_bytes(n, Ci, Co, T, V, 1)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    if wsb:
        assert lib.dsgcn_pwconv_wsplit(P(w), Ci, Co, P(ws), st) == 0
    wsp = P(ws) if wsb else None
    part = torch.zeros(lib.dsgcn_pwconv_partial_rows(n, Ci, Co, T, V, 1, 0), Co, 2, device=dev)
    ipart = torch.zeros(lib.dsgcn_pwconv_ipart_rows(n, Ci, Co, T, V, 1), Ci, 3, device=dev) if mode else None

    def fwd(q):
        assert lib.dsgcn_pwconv_fwd_ws(P(q['x1']), P(s1), P(h1), P(q['x2']), None, None, relu, P(w), P(b), P(q['z']), None,
                                       P(part), n, Ci, Co, T, V, 1, 0, 1, wsp, st) == 0

    def dgrad(q):
        assert lib.dsgcn_pwconv_dgrad_ws(P(q['x1']), P(s1), P(h1), P(q['x2']), None, None, relu, P(w), P(q['z']), None,
                                         P(q['gz']), None, P(A0), P(B0), P(q['dx']), P(q['dx2']), P(ipart), n, Ci, Co, T, V,
                                         1, 0, wsp, st) == 0
    return fwd, dgrad, part, ipart, ws


def check(keys, n):
    print(f'accuracy vs fp64 (n = {n}): z, stats(sum), stats(sumsq), dx, [dx2], [ipart u1]')
    for name, Ci, Co, T, mode in SHAPES:
        torch.manual_seed(Ci + Co)
        s1 = (torch.rand(Ci, device=dev) + .5) if mode else None
        h1 = (torch.randn(Ci, device=dev) * .1) if mode else None
        w = torch.randn(Co, Ci, device=dev) * Ci ** -.5; b = torch.randn(Co, device=dev)
        A0 = torch.randn(Co, device=dev) * 1e-1; B0 = torch.randn(Co, device=dev) * 1e-1
        q = operands(n, Ci, Co, T, mode, 5)
        v = q['x1'].double()
        if mode:
            v = v * s1.double().view(1, -1, 1, 1) + h1.double().view(1, -1, 1, 1)
            if mode == 2:
                v = v + q['x2'].double()
            pre = v
            v = v.clamp_min(0)
        zr = torch.einsum('oc,nctv->notv', w.double(), v) + b.double().view(1, -1, 1, 1)
        line = f'{name:8s} {Ci:4d}->{Co:4d} T={T:2d} m{mode}'
        for k in keys:
            fwd, dgrad, part, ipart, ws = run_form(k, n, Ci, Co, T, mode, (s1, h1, w, b, A0, B0), q)
            q['z'].fill_(float('nan')); q['dx'].fill_(float('nan'))
            if q['dx2'] is not None:
                q['dx2'].fill_(float('nan'))
            fwd(q); dgrad(q)
            torch.cuda.synchronize()
            dze = q['gz'].double() + A0.double().view(1, -1, 1, 1) + B0.double().view(1, -1, 1, 1) * q['z'].double()
            dv = torch.einsum('oc,notv->nctv', w.double(), dze)
            if mode:
                dv = dv * (pre > 0)
                dxr = dv * s1.double().view(1, -1, 1, 1)
            else:
                dxr = dv
            ps = part.double().sum(0)
            r = [rel(q['z'], zr), rel(ps[:, 0], zr.sum((0, 2, 3))), rel(ps[:, 1], (zr * zr).sum((0, 2, 3))), rel(q['dx'], dxr)]
            if mode == 2:
                r.append(rel(q['dx2'], dv))
            if mode:
                r.append(rel(ipart.double().sum(0)[:, 1], dv.sum((0, 2, 3))))
            ok = all(e < 6e-7 for e in r[:1] + r[3:4]) and all(e < 2e-5 for e in r)
            line += f' | k14={k}: ' + ' '.join(f'{e:.1e}' for e in r) + ('' if ok else '  <-- FAIL')
        print(line, flush=True)


def bench(keys, n, reps=20, nsets=4):
    print(f'HBM-cold launch times (us), n = {n}, {nsets} rotating operand sets')
    tot = {k: [0.0, 0.0] for k in keys}
    for name, Ci, Co, T, mode in SHAPES:
        torch.manual_seed(1)
        s1 = (torch.rand(Ci, device=dev) + .5) if mode else None
        h1 = (torch.randn(Ci, device=dev) * .1) if mode else None
        w = torch.randn(Co, Ci, device=dev) * Ci ** -.5; b = torch.zeros(Co, device=dev)
        A0 = torch.randn(Co, device=dev) * 1e-3; B0 = torch.randn(Co, device=dev) * 1e-3
        sets = [operands(n, Ci, Co, T, mode, 10 + i) for i in range(nsets)]
        for q in sets:
            q['z'].normal_()
        L = n * T * V
        nin = 2 if mode == 2 else 1
        line = f'{name:8s} {Ci:4d}->{Co:4d} T={T:2d} m{mode}'
        for k in keys:
            fwd, dgrad, part, ipart, ws = run_form(k, n, Ci, Co, T, mode, (s1, h1, w, b, A0, B0), sets[0])
            for j, (fn, nbytes) in enumerate(((fwd, 4 * L * (Ci * nin + Co)), (dgrad, 4 * L * (2 * Co + Ci * nin * (2 if mode else 1))))):
                for q in sets:
                    fn(q)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for r in range(reps):
                    fn(sets[r % nsets])
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / reps * 1e3
                tot[k][j] += us
                line += f' | k14={k} {"fwd" if j == 0 else "dgrad"} {us:6.1f} us {nbytes / us / 1e6:4.2f} TB/s {6 * 2.0 * Ci * Co * L / us / 1e6:5.0f} TF' if j == 0 else f' dgrad {us:6.1f} us {nbytes / us / 1e6:4.2f} TB/s'
        print(line, flush=True)
        if os.environ.get('KC_PHASES') and 2 in keys:
            import numpy as np
            fwd, dgrad, part, ipart, ws = run_form(2, n, Ci, Co, T, mode, (s1, h1, w, b, A0, B0), sets[0])
            for blk in (0, 300):
                assert lib.dsgcn_pwg2_phases_block(blk) == 0
                for nm, fn in (('fwd', fwd), ('dgrad', dgrad)):
                    torch.cuda.synchronize(); fn(sets[1]); torch.cuda.synchronize()
                    ph = np.zeros(64, dtype=np.int64)
                    assert lib.dsgcn_pwg2_phases(ph.ctypes.data) == 0
                    k_ = int(ph[63]); d = np.diff(ph[:k_]) / 100.0
                    if k_ < 6 or int(ph[0]) in _seen_stamps:      # (not a k_pwg3 launch / fewer workgroups than `blk`: stale stamps)
                        continue
                    _seen_stamps.add(int(ph[0]))
                    ch = d[2:-2].reshape(-1, 2)
                    print(f'    wg {blk} {nm}: table {d[0]:.2f} | chunk 0 in LDS {d[1]:.2f} | per chunk products/barrier: ' +
                          ' '.join(f'{a_:.2f}/{b_:.2f}' for a_, b_ in ch) + f' | drain {d[-2]:.2f} epilogue {d[-1]:.2f} | total {(ph[k_ - 1] - ph[0]) / 100.0:.1f} us')
            assert lib.dsgcn_pwg2_phases_block(0) == 0
    for k in keys:
        print(f'k14={k}: total fwd {tot[k][0]:.0f} us, dgrad {tot[k][1]:.0f} us')


def wgrad(n_check, n_bench, reps=20, nsets=4):
    """Weight gradient: k_wg2 (lab key 15 = 0) against k_wg3 (15 = 1): accuracy vs fp64 at n_check, HBM-cold times at n_bench."""
    print(f'weight gradient: relative L2 of dW, db vs fp64 (n = {n_check}) | HBM-cold us (n = {n_bench})')
    for name, Ci, Co, T, mode in SHAPES:
        torch.manual_seed(Ci * 3 + Co)
        s1 = (torch.rand(Ci, device=dev) + .5) if mode else None
        h1 = (torch.randn(Ci, device=dev) * .1) if mode else None
        A0 = torch.randn(Co, device=dev) * 1e-1; B0 = torch.randn(Co, device=dev) * 1e-1
        relu = 1 if mode else 0
        line = f'{name:8s} {Ci:4d}->{Co:4d} T={T:2d} m{mode}'
        for key in (0, 1):
            assert lib.dsgcn_pwconv_tuning(15, key) == 0
            for hasc in ((True,) if mode != 0 else (True, False)):
                res = []
                for n, timed in ((n_check, False), (n_bench, True)):
                    sets = [operands(n, Ci, Co, T, mode, 20 + i) for i in range(nsets if timed else 1)]
                    for q in sets:
                        q['z'].normal_()
                    splits = lib.dsgcn_pwconv_wgrad_splits(n, Ci, Co, T, V, 1)
                    pstride = Co * Ci + Co
                    wpart = torch.zeros(splits, pstride, device=dev)

                    def fn(q):
                        assert lib.dsgcn_pwconv_wgrad(P(q['x1']), P(s1), P(h1), P(q['x2']), None, None, relu, P(q['z']) if hasc else None,
                                                      None, P(q['gz']), None, P(A0) if hasc else None, P(B0) if hasc else None,
                                                      wpart.data_ptr(), wpart.data_ptr() + 4 * Co * Ci, pstride, n, Ci, Co, T, V, 1, 0, st) == 0
                    if not timed:
                        q = sets[0]
                        fn(q)
                        torch.cuda.synchronize()
                        v = q['x1'].double()
                        if mode:
                            v = v * s1.double().view(1, -1, 1, 1) + h1.double().view(1, -1, 1, 1)
                            if mode == 2:
                                v = v + q['x2'].double()
                            v = v.clamp_min(0)
                        dze = q['gz'].double()
                        if hasc:
                            dze = dze + A0.double().view(1, -1, 1, 1) + B0.double().view(1, -1, 1, 1) * q['z'].double()
                        dwr = torch.einsum('notv,nctv->oc', dze, v)
                        dw = wpart.double().sum(0)
                        e1, e2 = rel(dw[:Co * Ci].view(Co, Ci), dwr), rel(dw[Co * Ci:], dze.sum((0, 2, 3)))
                        res.append(f'{e1:.1e} {e2:.1e}' + ('' if e1 < 3e-7 and e2 < 2e-5 else ' <-- FAIL'))
                    else:
                        for q in sets:
                            fn(q)
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for r in range(reps):
                            fn(sets[r % nsets])
                        e1.record()
                        torch.cuda.synchronize()
                        us = e0.elapsed_time(e1) / reps * 1e3
                        nb = 4 * n * T * V * ((2 if hasc else 1) * Co + Ci * (2 if mode == 2 else 1))
                        res.append(f'{us:6.1f} us {nb / us / 1e6:4.2f} TB/s splits {splits}')
                line += f' | wg3={key}{"" if hasc else " noBN"}: ' + ' | '.join(res)
        print(line, flush=True)
    assert lib.dsgcn_pwconv_tuning(15, 1) == 0


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if a != 'wgrad']
    if 'wgrad' in sys.argv[1:]:
        wgrad(int(os.environ.get('KC_CHECK_N', 6)), int(os.environ.get('KC_N', 128)))
    if args or 'wgrad' not in sys.argv[1:]:
        keys = [int(a) for a in args] or [1, 2]
        check(keys, int(os.environ.get('KC_CHECK_N', 6)))
        bench(keys, int(os.environ.get('KC_N', 128)))
