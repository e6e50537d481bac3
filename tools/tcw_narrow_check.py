"""k_tcw (9-tap weight gradient, csrc/tcg.hip) with and without the wave mapping for co tiles of <= 64 rows (lab knob
dsgcn_tconv_tuning(1, v)): the partial rows must be the same BITS (every (co, ci, tap) keeps its one wave's walk over the
units) and the 64-channel layers faster.    python tools/tcw_narrow_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsgcn_amd import native
lib = native.lab_lib(); dev = 'cuda'; st = torch.cuda.current_stream().cuda_stream
P = lambda t: None if t is None else t.data_ptr()
n, V = 128, 25
# (name, Ci, Co, T, stride, KT)
SHAPES = [('s1', 64, 64, 64, 1, 9), ('s1k5', 64, 64, 64, 1, 5), ('s1k3', 64, 48, 64, 1, 3), ('s2t', 64, 128, 64, 2, 9),
          ('s2', 128, 128, 32, 1, 9), ('192', 64, 192, 32, 1, 9), ('s1s2', 64, 64, 64, 2, 9), ('s3', 256, 256, 16, 1, 9)]


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


bad = 0
for name, Ci, Co, T, s, KT in SHAPES:
    g = torch.Generator().manual_seed(Ci + Co + T)
    To = (T + s - 1) // s
    x1 = torch.randn(n, Ci, T, V, generator=g).to(dev); x2 = torch.randn(n, Ci, T, V, generator=g).to(dev)
    s1 = (torch.rand(Ci, generator=g) + .5).to(dev); h1 = (torch.randn(Ci, generator=g) * .1).to(dev)
    z = torch.randn(n, Co, To, V, generator=g).to(dev); gz = torch.randn(n, Co, To, V, generator=g).to(dev)
    A0 = (torch.randn(Co, generator=g) * 1e-3).to(dev); B0 = (torch.randn(Co, generator=g) * 1e-3).to(dev)
    splits = lib.dsgcn_tconv_wgrad_splits(n, Ci, Co, T, V, KT, s)
    assert splits > 0, name
    pstride = Co * Ci * KT + Co
    res, tm = {}, {}
    for mode in (0, 1):
        assert lib.dsgcn_tconv_tuning(1, mode) == 0
        wp = torch.full((splits, pstride), float('nan'), device=dev)
        fn = lambda: lib.dsgcn_tconv_wgrad(P(x1), P(s1), P(h1), P(x2), None, None, 1, P(z), P(gz), P(A0), P(B0), wp.data_ptr(),
                                           wp.data_ptr() + 4 * Co * Ci * KT, pstride, n, Ci, Co, T, V, KT, s, st)
        assert fn() == 0
        torch.cuda.synchronize()
        res[mode] = wp.clone()
    for rnd in range(3):                               # timing: the two forms interleaved, best of three
        for mode in (1, 0):
            assert lib.dsgcn_tconv_tuning(1, mode) == 0
            t = timeit(fn)
            tm[mode] = min(tm.get(mode, 1e9), t)
    same = bool((res[0].view(torch.int32) == res[1].view(torch.int32)).all())
    bad += not same
    print(f'{name:5s} {Ci:3d}->{Co:3d} T={T:2d} stride {s} taps {KT}: {tm[0]:7.1f} -> {tm[1]:7.1f} us, same bits {same}, finite {bool(torch.isfinite(res[1]).all())}')
print('ALL BIT-IDENTICAL' if not bad else f'{bad} SHAPES DIFFER')
sys.exit(1 if bad else 0)
