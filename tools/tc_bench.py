"""Per-layer GPU time of the K-D tapconv launches (fwd / dgrad / wgrad), C ABI called directly: python tools/tc_bench.py
(TC_ONLY=conv|elem: only the conv windows / only the max-pool and pass-through windows)."""
import os, sys, ctypes as ct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsgcn_amd import kernels as K, native
lib = native.lib()
dev = 'cuda'
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
cfg3 = [(3, 1), (3, 2), (3, 3), (3, 4), ('max', 3), '1x1']
cfg5 = [(5, 1), (5, 2), ('max', 3), '1x1']
pts = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]
for name, cfg, C, T, V1, stride in [('ds', cfg3, 64, 64, 26, 1), ('ds', cfg3, 128, 64, 26, 2), ('ds', cfg3, 128, 32, 26, 1),
                                    ('ds', cfg3, 256, 16, 26, 1), ('ctr', cfg5, 64, 64, 25, 1), ('ctr', cfg5, 128, 32, 25, 1),
                                    ('ctr', cfg5, 256, 16, 25, 1)]:
    nb = len(cfg); mid = C // nb
    widths = [C - mid * (nb - 1)] + [mid] * (nb - 1) if name == 'ds' else [C // 4] * 4
    ks = cfg[0][0]
    wl = [torch.randn(w, w, ks, 1, device=dev) * .1 for w in widths]; bl = [torch.randn(w, device=dev) for w in widths]
    KT, types, c0s, bcs, dils, ws, bs = K._branch_tables(cfg, widths, wl, bl)
    only = os.environ.get('TC_ONLY')            # conv | elem: time a subset of the windows
    if only:
        keep = [i for i, t in enumerate(types) if (t == 0) == (only == 'conv')]
        types, c0s, bcs, dils, ws, bs = [[x[i] for i in keep] for x in (types, c0s, bcs, dils, ws, bs)]
    n = 128
    Tout = (T + stride - 1) // stride
    h = torch.randn(n, C, T, V1, device=dev); o = torch.empty(n, C, Tout, V1, device=dev); go = torch.randn_like(o); dh = torch.empty_like(h)
    tabs = [K._int_array(x) for x in (types, c0s, c0s, bcs, bcs, dils)]
    nbr = len(types)
    wp, bp = K._ptr_array(ws), K._ptr_array(bs)
    offs, off = [], 0
    for t, bc in zip(types, bcs):
        offs.append(off)
        if t == 0: off += bc * bc * KT + bc
    pstride = max(off, 1)
    splits = max(64, min(1024, (1 << 21) // pstride)) // 4 * 4 if max(bcs) <= 32 else max(16, min(256, (1 << 21) // pstride))
    pref = lib.dsgcn_tapconv_wgrad_splits(n, C, C, T, V1, stride, KT, nbr, tabs[0], tabs[3], tabs[4], tabs[5])
    splits = pref or splits
    part = torch.empty(splits, pstride, device=dev)
    base = part.data_ptr()
    dwp = (ct.c_void_p * nbr)(*[base + 4 * o_ if t == 0 else None for t, o_ in zip(types, offs)])
    dbp = (ct.c_void_p * nbr)(*[base + 4 * (o_ + bc * bc * KT) if t == 0 else None for t, o_, bc in zip(types, offs, bcs)])
    def fwd(): assert lib.dsgcn_tapconv_fwd(h.data_ptr(), o.data_ptr(), n, C, C, T, V1, stride, KT, nbr, *tabs, wp, bp, st) == 0
    def dgr(): assert lib.dsgcn_tapconv_dgrad(h.data_ptr(), go.data_ptr(), dh.data_ptr(), n, C, C, T, V1, stride, KT, nbr, *tabs, wp, st) == 0
    def wgr(): assert lib.dsgcn_tapconv_wgrad(h.data_ptr(), go.data_ptr(), n, C, C, T, V1, stride, KT, nbr, *tabs, dwp, dbp, splits, pstride, st) == 0
    mb = (h.numel() + o.numel()) * 4 / 1e6
    line = f'{name} C={C:3d} T={T} s={stride} ({mb:5.1f} MB): wgrad {timeit(wgr):6.1f} '
    line += f'| fwd {timeit(fwd):6.1f} dgrad {timeit(dgr):6.1f} '
    print(line, flush=True)
