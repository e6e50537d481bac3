"""K-A' with the SHARED adjacency (ST-GCN's unit_gcn; csrc/aggsum.hip): the adjacency's MFMA fragments in registers (lab key
3 = 1) against the per-(unit, subset) reload (0), and the launch geometry of both directions (keys 1 / 2), HIP-event timed
over ST-GCN's layer shapes.   python tools/kap_shared.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsgcn_amd import native
lib = native.lab_lib(); dev = 'cuda'; st = torch.cuda.current_stream().cuda_stream
n, V, K = 128, 25, 3
SHAPES = [(64, 64), (128, 32), (256, 16)]


def timeit(fn, reps=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


bufs = {}
for Co, T in SHAPES:
    p = torch.randn(n, K * Co, T, V, device=dev); A = torch.randn(K, V, V, device=dev) * 0.2
    rows = lib.dsgcn_aggsum_bwd_piece_rows(n, K, Co, T, V)
    bufs[(Co, T)] = dict(p=p, A=A, y=torch.empty(n, Co, T, V, device=dev), gy=torch.randn(n, Co, T, V, device=dev),
                         dp=torch.empty_like(p), piece=torch.empty(max(rows, n * Co, 16384), K, V, V, device=dev),
                         part=torch.empty(lib.dsgcn_aggsum_partial_rows(n, T, V), Co, 2, device=dev))


def line(tag):
    row = []
    for Co, T in SHAPES:
        b = bufs[(Co, T)]
        fw = lambda: lib.dsgcn_aggsum_fwd(b['p'].data_ptr(), b['A'].data_ptr(), 0, V * V, 0, b['y'].data_ptr(), b['part'].data_ptr(),
                                          n, K, Co, T, V, st)
        bw = lambda: lib.dsgcn_aggsum_bwd(b['p'].data_ptr(), b['A'].data_ptr(), 0, V * V, 0, b['gy'].data_ptr(), None, None, None,
                                          b['dp'].data_ptr(), b['piece'].data_ptr(), Co * K * V * V, V * V, K * V * V, n, K, Co, T, V, st)
        tf, tb = timeit(fw), timeit(bw)
        bf, bb = 4 * n * Co * (K + 1) * T * V, 4 * n * Co * (2 * K + 1) * T * V
        row.append(f'{Co:3d}x{T:2d} fwd {tf:6.1f}us {bf / tf / 1e6:4.2f}TB/s  bwd {tb:6.1f}us {bb / tb / 1e6:4.2f}TB/s')
    print(f'{tag:22s} ' + ' | '.join(row), flush=True)


for reg in (0, 1):
    assert lib.dsgcn_aggsum_tuning(3, reg) == 0
    line(f'registers={reg}')
assert lib.dsgcn_aggsum_tuning(3, 1) == 0
for w in (4096, 8192):
    assert lib.dsgcn_aggsum_tuning(1, w) == 0
    line(f'fwd waves {w}')
assert lib.dsgcn_aggsum_tuning(1, 0) == 0
for w in (1024, 1536, 2048, 2560, 3072, 4096):
    assert lib.dsgcn_aggsum_tuning(2, w) == 0
    line(f'bwd workgroups {w}')
assert lib.dsgcn_aggsum_tuning(2, 0) == 0
