"""K-A' (csrc/aggsum.hip) launch-geometry sweep over CTR-GCN's layer mix: persistent workgroups of the backward (lab key 2)
and waves of the forward (lab key 1), HIP-event timed per shape class.   python tools/kap_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsgcn_amd import native
lib = native.lab_lib(); dev = 'cuda'; st = torch.cuda.current_stream().cuda_stream
n, V, K = 128, 25, 3
SHAPES = [(64, 64), (128, 64), (128, 32), (256, 32), (256, 16)]


def timeit(fn, reps=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


bufs = {}
for Co, T in SHAPES:
    p = torch.randn(n, K * Co, T, V, device=dev); ah = torch.randn(n, K * Co, V, V, device=dev) * 0.2
    bufs[(Co, T)] = dict(p=p, ah=ah, y=torch.empty(n, Co, T, V, device=dev), gy=torch.randn(n, Co, T, V, device=dev),
                         dp=torch.empty_like(p), dah=torch.empty_like(ah),
                         part=torch.empty(lib.dsgcn_aggsum_partial_rows(n, T, V), Co, 2, device=dev))
for key, vals in ((2, [0, 1024, 1536, 2048, 2560, 3072, 4096]), (1, [0, 2048, 3072, 4096, 6144])):
    for val in vals:
        assert lib.dsgcn_aggsum_tuning(key, val) == 0
        row = []
        for Co, T in SHAPES:
            b = bufs[(Co, T)]
            if key == 2:
                fn = lambda: lib.dsgcn_aggsum_bwd(b['p'].data_ptr(), b['ah'].data_ptr(), K * Co * V * V, Co * V * V, V * V, b['gy'].data_ptr(),
                                                  None, None, None, b['dp'].data_ptr(), b['dah'].data_ptr(), K * Co * V * V, Co * V * V, V * V,
                                                  n, K, Co, T, V, st)
                nbytes = 4 * n * Co * ((2 * K + 1) * T * V + 2 * K * V * V)
            else:
                fn = lambda: lib.dsgcn_aggsum_fwd(b['p'].data_ptr(), b['ah'].data_ptr(), K * Co * V * V, Co * V * V, V * V, b['y'].data_ptr(),
                                                  b['part'].data_ptr(), n, K, Co, T, V, st)
                nbytes = 4 * n * Co * ((K + 1) * T * V + K * V * V)
            t = timeit(fn)
            row.append(f'{Co:3d}x{T:2d} {t:6.1f}us {nbytes / t / 1e6:4.2f}TB/s')
        print(('bwd wgs ' if key == 2 else 'fwd waves ') + f'{val:5d}: ' + ' | '.join(row), flush=True)
    assert lib.dsgcn_aggsum_tuning(key, 0) == 0
