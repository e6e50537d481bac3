"""The temporal stage of a dgmstcn unit (between its two 1x1 convs) at the DS-STGCN layer shapes, staged form (branch_act ->
tapconv -> combine) against the split layout (csrc/tmsplit.hip): forward and backward time per layer (HIP events over `reps`
repetitions on rotating operands larger than the Infinity Cache would need too much memory here: the planes are L2-cold but
MALL-warm, as in the step) and the agreement of every output / gradient between the two.
    python tools/tsp_bench.py [n=128] [reps=20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsgcn_amd import kernels as K, native

if os.environ.get('TS_EXP'):
    native._lib = native.lab_lib()
    assert native._lib.dsgcn_tms_split_tuning(0, int(os.environ['TS_EXP'])) == 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
STRIDE = int(os.environ.get('TS_STRIDE', 1))      # 2: the two stride-2 units of the network (128 ch: T 64 -> 32, 256 ch: T 32 -> 16)
dev = torch.device('cuda')
cfg = [(3, 1), (3, 2), (3, 3), (3, 4), ('max', 3), '1x1']


def make(C, T, V, seed=0):
    g = torch.Generator().manual_seed(seed)
    mid = C // 6
    widths = [C - 5 * mid] + [mid] * 5
    n_act = C - mid
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
    d = dict(z=r(n, C, T, V), zaug=r(n, C, T), scale=torch.cat([torch.rand(n_act, generator=g) + 0.5, torch.ones(mid)]).to(dev),
             shift=torch.cat([torch.randn(n_act, generator=g) * 0.3, torch.zeros(mid)]).to(dev),
             cw=[r(w, w, 3, 1, scale=(3 * w) ** -0.5) for w in widths[:4]], cb=[r(w, scale=0.1) for w in widths[:4]],
             coeff=r(25, scale=0.5), gamma=(torch.rand(C, generator=g) + 0.5).to(dev), beta=r(C, scale=0.2),
             gf=r(n, C, T // STRIDE, V), gsc=r(C), gsh=r(C), widths=widths, n_act=n_act)
    return d


def run(d, split, time_it):
    K.SPLIT_TEMPORAL = '2' if split else '0'
    K.FUSED_TEMPORAL = '0'
    leaves = {k: d[k].clone().requires_grad_() for k in ('z', 'zaug', 'scale', 'shift', 'coeff', 'gamma', 'beta')}
    tw = [w.clone().requires_grad_() for w in d['cw']]
    tb = [b.clone().requires_grad_() for b in d['cb']]

    def fwd():
        return K.temporal_ms(leaves['z'], leaves['zaug'], leaves['scale'], leaves['shift'], d['n_act'], cfg, d['widths'], tw, tb,
                             leaves['coeff'], STRIDE, leaves['gamma'], leaves['beta'], 1e-5, True)

    def loss(out):
        f, sc, sh = out[0], out[1], out[2]
        return (f * d['gf']).sum() + (sc * d['gsc']).sum() + (sh * d['gsh']).sum()

    out = fwd()
    loss(out).backward()
    res = dict(f=out[0], sc=out[1], sh=out[2], mean=out[3], var=out[4])
    for k, t in leaves.items():
        res['d' + k] = t.grad.clone()
    for i in range(4):
        res[f'dw{i}'], res[f'db{i}'] = tw[i].grad.clone(), tb[i].grad.clone()
    tf = tb_ = None
    if time_it:
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb_ = 0.0
        for _ in range(reps):
            for t in list(leaves.values()) + tw + tb:
                t.grad = None
            e[0].record()
            out = fwd()
            l = loss(out)
            e[1].record()
            l.backward()
            e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1])
            tb_ += e[1].elapsed_time(e[2])
        tf, tb_ = tf / reps * 1e3, tb_ / reps * 1e3
    return res, tf, tb_


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


for C, T in (((64, 64), (128, 32), (256, 16)) if STRIDE == 1 else ((128, 64), (256, 32))):
    d = make(C, T, 25, seed=C)
    ok = native.lib().dsgcn_tms_split_rows(-1, n, C, T, 25, STRIDE, 3, 6, K._int_array([0, 0, 0, 0, 1, 2]),
                                           K._int_array([sum(d['widths'][:i]) for i in range(6)]), K._int_array(d['widths']),
                                           K._int_array([1, 2, 3, 4, 1, 1]))
    r0, f0, b0 = run(d, False, True)
    r1, f1, b1 = run(d, True, True)
    worst = max((rel(r1[k], r0[k]), k) for k in r0)
    if os.environ.get('TS_PHASES'):
        import numpy as np
        ph = np.zeros(64, dtype=np.int64)
        assert native._lib.dsgcn_tms_split_phases(0, ph.ctypes.data) == 0
        k = int(ph[63])
        d_ = np.diff(ph[:k]) * 10e-3
        print(f'  k_tspw stamps (us): start->cleared {d_[0]:.2f}; per unit [commit+barrier, issue, products+barrier]: ' +
              ' '.join(f'[{d_[i]:.2f} {d_[i + 1]:.2f} {d_[i + 2]:.2f}]' for i in range(1, k - 2, 3)) + f'; tail {d_[k - 2]:.2f}')
    if os.environ.get('TS_CONV_BLOCK'):
        import numpy as np
        for blk in [int(x) for x in os.environ['TS_CONV_BLOCK'].split(',')]:
            assert native._lib.dsgcn_tms_split_tuning(1, blk) == 0
            for split_dir in ('fwd', 'bwd'):
                r_, _, _ = run(d, True, False)      # fwd then bwd launch: the stamps are of the LAST k_tsp launch = backward
                ph = np.zeros(16, dtype=np.int64)
                if split_dir == 'fwd':
                    K.SPLIT_TEMPORAL = '2'; K.FUSED_TEMPORAL = '0'
                    with torch.no_grad():
                        K.temporal_ms(d['z'], d['zaug'], d['scale'], d['shift'], d['n_act'], cfg, d['widths'], d['cw'], d['cb'],
                                      d['coeff'], 1, d['gamma'], d['beta'], 1e-5, True)
                torch.cuda.synchronize()
                assert native._lib.dsgcn_tms_split_phases(1, ph.ctypes.data) == 0
                k = int(ph[15])
                print(f'  k_tsp {split_dir} conv block {blk} (us): ' + ' '.join(f'{x:.2f}' for x in np.diff(ph[:k]) * 10e-3) +
                      f'  total {(ph[k - 1] - ph[0]) * 10e-3:.2f}')
    print(f'C={C:3d} T={T:2d} eligible={ok}  staged fwd {f0:7.1f} bwd {b0:7.1f} us   split fwd {f1:7.1f} bwd {b1:7.1f} us   '
          f'max rel diff {worst[0]:.2e} ({worst[1]})', flush=True)
