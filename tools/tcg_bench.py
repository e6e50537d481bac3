"""Dense temporal conv (unit_tcn's 9-tap conv) per ST-GCN layer shape: forward / data gradient / weight gradient of the
GEMM form (csrc/tcg.hip) next to the first-generation tap kernels, HIP-event timed through the C ABI, 128 person-samples.
    python tools/tcg_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as ct
import torch
from dsgcn_amd import native
LAB = os.environ.get('TC_LAB') == '1'
lib = native.lab_lib() if LAB else native.lib(); dev = 'cuda'; st = torch.cuda.current_stream().cuda_stream
if LAB and os.environ.get('TC_QUAD') is not None:
    assert lib.dsgcn_tconv_tuning(0, int(os.environ['TC_QUAD'])) == 0      # 0: the 4-byte staging of k_tcg (round 3)
n, V, KT = int(os.environ.get('TC_N', 128)), 25, 9
SHAPES = [('s1', 64, 64, 64, 1), ('s2t', 64, 128, 64, 2), ('s2', 128, 128, 32, 1), ('s3t', 128, 256, 32, 2), ('s3', 256, 256, 16, 1)]
P = lambda t: None if t is None else t.data_ptr()


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def ia(v):
    return (ct.c_int * len(v))(*v)


print(f'{"layer":5s} {"Ci":>4s} {"Co":>4s} {"T":>3s} s | GEMM fwd  dgrad  wgrad (us; TF of fp32-equivalent products) | first-gen fwd  dgrad  wgrad')
for name, Ci, Co, T, s in SHAPES:
    To = (T + s - 1) // s
    x1 = torch.randn(n, Ci, T, V, device=dev); x2 = torch.randn(n, Ci, T, V, device=dev)
    s1 = torch.rand(Ci, device=dev) + .5; h1 = torch.randn(Ci, device=dev) * .1
    w = torch.randn(Co, Ci, KT, 1, device=dev) * (Ci * KT) ** -.5; b = torch.zeros(Co, device=dev)
    z = torch.empty(n, Co, To, V, device=dev); gz = torch.randn(n, Co, To, V, device=dev)
    A0 = torch.randn(Co, device=dev) * 1e-3; B0 = torch.randn(Co, device=dev) * 1e-3
    dx1 = torch.empty_like(x1); dx2 = torch.empty_like(x1)
    wsb = lib.dsgcn_tconv_ws_bytes(n, Ci, Co, T, V, KT, s)
    assert wsb
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    assert lib.dsgcn_tconv_wsplit(P(w), Ci, Co, KT, P(ws), st) == 0
    part = torch.empty(lib.dsgcn_tconv_rows(0, n, Ci, Co, T, V, KT, s), Co, 2, device=dev)
    ipart = torch.empty(lib.dsgcn_tconv_rows(1, n, Ci, Co, T, V, KT, s), Ci, 3, device=dev)
    splits = lib.dsgcn_tconv_wgrad_splits(n, Ci, Co, T, V, KT, s)
    pstride = Co * Ci * KT + Co
    wp = torch.empty(splits, pstride, device=dev)
    f = lambda: lib.dsgcn_tconv_fwd(P(x1), P(s1), P(h1), P(x2), None, None, 1, P(ws), P(b), P(z), P(part), n, Ci, Co, T, V, KT, s, st)
    d = lambda: lib.dsgcn_tconv_dgrad(P(x1), P(s1), P(h1), P(x2), None, None, 1, P(ws), P(z), P(gz), P(A0), P(B0), P(dx1), P(dx2), P(ipart), n, Ci, Co, T, V, KT, s, st)
    g = lambda: lib.dsgcn_tconv_wgrad(P(x1), P(s1), P(h1), P(x2), None, None, 1, P(z), P(gz), P(A0), P(B0), wp.data_ptr(), wp.data_ptr() + 4 * Co * Ci * KT, pstride, n, Ci, Co, T, V, KT, s, st)
    assert f() == 0 and d() == 0 and g() == 0
    # first generation: materialised operands, one conv window
    tabs = (ia([0]), ia([0]), ia([0]), ia([Ci]), ia([Co]), ia([1]))
    wsp = (ct.c_void_p * 1)(w.data_ptr()); bsp = (ct.c_void_p * 1)(b.data_ptr())
    f0 = lambda: lib.dsgcn_tapconv_fwd(P(x1), P(z), n, Ci, Co, T, V, s, KT, 1, *tabs, wsp, bsp, st)
    d0 = lambda: lib.dsgcn_tapconv_dgrad(P(x1), P(gz), P(dx1), n, Ci, Co, T, V, s, KT, 1, *tabs, wsp, st)
    sp0 = 64
    wp0 = torch.empty(sp0, pstride, device=dev)
    dwp = (ct.c_void_p * 1)(wp0.data_ptr()); dbp = (ct.c_void_p * 1)(wp0.data_ptr() + 4 * Co * Ci * KT)
    g0 = lambda: lib.dsgcn_tapconv_wgrad(P(x1), P(gz), n, Ci, Co, T, V, s, KT, 1, *tabs, dwp, dbp, sp0, pstride, st)
    fl = 2.0 * n * To * V * Ci * Co * KT
    r = [timeit(k) for k in (f, d, g)]
    try:
        if os.environ.get('TC_SKIP_GEN1'):
            raise RuntimeError('skipped')
        assert f0() == 0 and d0() == 0 and g0() == 0
        r0 = [timeit(k) for k in (f0, d0, g0)]
    except Exception as e:
        r0 = [float('nan')] * 3
    print(f'{name:5s} {Ci:4d} {Co:4d} {T:3d} {s} | ' + '  '.join(f'{t:6.0f} ({fl / t / 1e6:5.0f})' for t in r) + ' | ' +
          '  '.join(f'{t:6.0f}' for t in r0) + f'   splits {splits}', flush=True)
    if LAB:
        import numpy as np
        torch.cuda.synchronize(); g(); torch.cuda.synchronize()
        ph = np.zeros(64, dtype=np.int64)
        assert lib.dsgcn_tcw_phases(ph.ctypes.data) == 0
        k_ = int(ph[63]) // 4 * 4
        d = np.diff(ph[:k_ + 1])[:k_ - 1] / 100.0 if k_ > 4 else []
        rows = [(ph[4 * i + 1] - ph[4 * i], ph[4 * i + 2] - ph[4 * i + 1], ph[4 * i + 3] - ph[4 * i + 2],
                 (ph[4 * i + 4] - ph[4 * i + 3]) if 4 * i + 4 < k_ else 0) for i in range(k_ // 4)]
        print('    k_tcw tap-group steps (us: issue / barrier wait / products / commit+loop):  ' +
              '  '.join('/'.join(f'{x / 100.0:.2f}' for x in r) for r in rows[:9]))

