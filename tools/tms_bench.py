"""Per-layer GPU time of the fused temporal stage (csrc/tms.hip: fwd / dgrad / wgrad) through the C ABI, next to the staged
chain it replaces (branch_act + tapconv + combine), cold operands (rotating sets):  python tools/tms_bench.py
TMS_ONLY=conv|elem: only the conv windows / only the max-pool and pass-through windows."""
import ctypes as ct
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dsgcn_amd import kernels as K, native  # noqa: E402

lib = native.lab_lib() if os.environ.get('TMS_LAB') else native.lib()
if os.environ.get('TMS_LAB'):
    for kv in os.environ['TMS_LAB'].split(','):
        k_, v_ = kv.split(':')
        lib.dsgcn_tms_tuning(int(k_), int(v_))
dev = 'cuda'
st = torch.cuda.current_stream().cuda_stream
P = lambda t: None if t is None else t.data_ptr()  # noqa: E731


def timeit(fn, nsets, reps=24):
    for i in range(nsets):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        fn(r % nsets)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cfg3 = [(3, 1), (3, 2), (3, 3), (3, 4), ('max', 3), '1x1']
cfg5 = [(5, 1), (5, 2), ('max', 3), '1x1']
n = int(os.environ.get('TMS_N', '128'))
NS = 4
for name, cfg, C, T, V, stride, aug in [('ds', cfg3, 64, 64, 25, 1, 1), ('ds', cfg3, 128, 64, 25, 2, 1), ('ds', cfg3, 128, 32, 25, 1, 1),
                                        ('ds', cfg3, 256, 32, 25, 2, 1), ('ds', cfg3, 256, 16, 25, 1, 1),
                                        ('ctr', cfg5, 64, 64, 25, 1, 0), ('ctr', cfg5, 256, 16, 25, 1, 0)]:
    nb = len(cfg)
    mid = C // nb
    widths = [C - mid * (nb - 1)] + [mid] * (nb - 1) if name == 'ds' else [C // 4] * 4
    ks = cfg[0][0]
    wl = [torch.randn(w, w, ks, 1, device=dev) * .1 for w in widths]
    bl = [torch.randn(w, device=dev) for w in widths]
    KT, types, c0s, bcs, dils, ws, bs = K._branch_tables(cfg, widths, wl, bl)
    only = os.environ.get('TMS_ONLY')
    if only:
        keep = [i for i, t in enumerate(types) if (t == 0) == (only == 'conv')]
        types, c0s, bcs, dils, ws, bs = [[x[i] for i in keep] for x in (types, c0s, bcs, dils, ws, bs)]
    nbr = len(types)
    Tout = (T + stride - 1) // stride
    n_act = C - (widths[-1] if name == 'ds' else C // 4)
    sets = [dict(z=torch.randn(n, C, T, V, device=dev), zaug=torch.randn(n, C, T, device=dev) if aug else None,
                 f=torch.empty(n, C, Tout, V, device=dev), oaug=torch.empty(n, C, Tout, device=dev) if aug else None,
                 gf=torch.randn(n, C, Tout, V, device=dev), dz=torch.empty(n, C, T, V, device=dev),
                 dzaug=torch.empty(n, C, T, device=dev) if aug else None) for _ in range(NS)]
    scale = torch.rand(C, device=dev) + .5
    shift = torch.randn(C, device=dev) * .1
    coeff = torch.randn(V, device=dev) * .5 if aug else None
    A0 = torch.randn(C, device=dev) * 1e-3
    B0 = torch.randn(C, device=dev) * 1e-3
    tabs = [K._int_array(x) for x in (types, c0s, bcs, dils)]
    rows = [lib.dsgcn_tms_rows(w_, n, C, T, V, stride, KT, nbr, tabs[0], tabs[2], tabs[3], aug) for w_ in range(3)]
    if rows[0] <= 0:
        print(name, C, T, 'not eligible')
        continue
    stats = torch.empty(rows[0], C, 2, device=dev)
    paff = torch.empty(rows[1], C, 2, device=dev)
    pcoeff = torch.empty(rows[1] * nbr, V, device=dev) if aug else None
    offs, off = [], 0
    for t, bc in zip(types, bcs):
        offs.append(off)
        if t == 0:
            off += bc * bc * KT + bc
    pstride = max(off, 1)
    part = torch.empty(max(rows[2], 1), pstride, device=dev)
    base = part.data_ptr()
    dwp = (ct.c_void_p * nbr)(*[base + 4 * o_ if t == 0 else None for t, o_ in zip(types, offs)])
    dbp = (ct.c_void_p * nbr)(*[base + 4 * (o_ + bc * bc * KT) if t == 0 else None for t, o_, bc in zip(types, offs, bcs)])
    wp, bp = K._ptr_array(ws), K._ptr_array(bs)

    def fwd(i):
        q = sets[i]
        assert lib.dsgcn_tms_fwd(P(q['z']), P(q['zaug']), P(scale), P(shift), n_act, P(coeff), P(q['f']), P(q['oaug']), P(stats),
                                 n, C, T, V, stride, KT, nbr, *tabs, wp, bp, st) == 0

    def dgr(i):
        q = sets[i]
        assert lib.dsgcn_tms_dgrad(P(q['z']), P(q['zaug']), P(scale), P(shift), n_act, P(coeff), P(q['gf']), P(q['f']), P(A0),
                                   P(B0), P(q['oaug']), P(q['dz']), P(q['dzaug']), P(paff), P(pcoeff), n, C, T, V, stride, KT,
                                   nbr, *tabs, wp, st) == 0

    def wgr(i):
        q = sets[i]
        assert lib.dsgcn_tms_wgrad(P(q['z']), P(q['zaug']), P(scale), P(shift), n_act, P(coeff), P(q['gf']), P(q['f']), P(A0),
                                   P(B0), n, C, T, V, stride, KT, nbr, *tabs, dwp, dbp, pstride, st) == 0
    mb_f = (sets[0]['z'].numel() + sets[0]['f'].numel()) * 4 / 1e6
    mb_d = (2 * sets[0]['z'].numel() + 2 * sets[0]['f'].numel()) * 4 / 1e6
    tf, td = timeit(fwd, NS), timeit(dgr, NS)
    tw = timeit(wgr, NS) if 0 in types else 0.0
    print(f'{name} C={C:3d} T={T:3d} s={stride} rows {rows}: fwd {tf:6.1f} us ({mb_f / tf * 1e-3:4.2f} TB/s)  dgrad {td:6.1f} us '
          f'({mb_d / td * 1e-3:4.2f} TB/s)  wgrad {tw:6.1f} us', flush=True)
