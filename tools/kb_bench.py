"""K-B alone: forward / backward time of the dynamic-adjacency kernels per DS-STGCN block (128 samples), C ABI timed with
HIP events, nothing else on the GPU:  python tools/kb_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dsgcn_amd as D
from dsgcn_amd import native
from bench import ds_cfg
LAB = os.environ.get('KB_LAB') == '1'      # KB_LAB=1: the lab build, prints the backward's phase times
lib = native.lab_lib() if LAB else native.lib(); st = torch.cuda.current_stream().cuda_stream
np.random.seed(0); torch.manual_seed(0)
m = D.build_model(ds_cfg(60)).cuda()
n, V = int(os.environ.get("KB_N", 64)), 25
P_ = lambda t: t.data_ptr()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


seen = set()
for blk in m.backbone.gcn:
    g = blk.gcn
    mid = g.edge_linears.weight.shape[1]
    if mid in seen:
        continue
    seen.add(mid)
    we, be = g.edge_linears.weight.flatten(1).contiguous(), g.edge_linears.bias
    E = we.shape[0] // mid
    Pn = g.conv1_se.weight.shape[0] // mid
    R, ld = (4 + Pn) * mid, 32
    proj = torch.randn(n, R, ld, device='cuda')
    alpha, beta = torch.randn(3, device='cuda') * .5, torch.randn(3, device='cuda') * .5
    ahat = torch.empty(n, 3 * mid, V, V, device='cuda'); dahat = torch.randn_like(ahat)
    dd = torch.empty_like(ahat); dproj = torch.empty_like(proj)
    ps = lib.dsgcn_dynadj_partial_stride(mid, V, E)
    ppar = torch.empty(n, ps, device='cuda')
    nt, et = g.node_type_idx, g.edge_type_idx
    f = lambda: lib.dsgcn_dynadj_fwd(P_(proj), P_(g.A), P_(alpha), P_(beta), P_(we), P_(be), P_(nt), P_(et), P_(ahat), n, mid, V, ld, Pn, E, st)
    b = lambda: lib.dsgcn_dynadj_bwd(P_(proj), P_(alpha), P_(beta), P_(we), P_(be), P_(nt), P_(et), P_(dahat), P_(dd), P_(dproj), P_(ppar), ps, n, mid, V, ld, Pn, E, st)
    assert f() == 0 and b() == 0
    print(f'mid={mid:3d} E={E} P={Pn}: fwd {timeit(f):6.1f} us  bwd {timeit(b):6.1f} us   (Ahat {ahat.numel()*4/1e6:.1f} MB)')
    if LAB:
        torch.cuda.synchronize(); b(); torch.cuda.synchronize()
        ph = np.zeros((3, 8), dtype=np.int64)
        assert lib.dsgcn_dynadj_phases(ph.ctypes.data) == 0
        names = ['prepare', 'pass1', 'sums+masked', 'gram', 'gemms', 'dproj']
        for k in range(3):
            d = np.diff(ph[k, :7]) / 100.0
            if k != 1:      # no stamp 4 on the plain subsets
                d = np.array([d[0], d[1], d[2], (ph[k, 5] - ph[k, 3]) / 100.0, 0.0, d[5]])
            print(f'   subset {k}: ' + '  '.join(f'{nm} {x:5.1f}' for nm, x in zip(names, d)) + f'   total {(ph[k, 6] - ph[k, 0]) / 100.0:.1f} us')

