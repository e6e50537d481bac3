"""Per-launch time and HBM rate of the elementwise stages of a DS block at the three stage shapes (n = 128):
    python tools/ew_bench.py [reps]
Launches rotate over EW_SETS (default 8) independent operand sets, so nothing is found in the 256 MB infinity cache
(EW_SETS=1: everything cache-resident).  GB/s counts every operand plane once (algorithmic bytes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace as NS
from dsgcn_amd import native
lib = native.lib(); dev = 'cuda'; st = torch.cuda.current_stream().cuda_stream
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 48
nsets = int(os.environ.get('EW_SETS', '8'))
n, V = 128, 25
P = lambda t: t.data_ptr() if t is not None else None


def timeit(fn):
    for i in range(nsets):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fn(i % nsets)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for (C, T) in [(64, 64), (128, 32), (256, 16)]:
    L = T * V
    pl = n * C * L * 4 / 1e6            # MB of one V-layout plane set
    pl1 = n * C * T * (V + 1) * 4 / 1e6
    sc = torch.rand(C, device=dev) + .5; sh = torch.randn(C, device=dev) * .1
    A0 = torch.randn(C, device=dev) * 1e-3; B0 = torch.randn(C, device=dev) * 1e-3
    part2 = torch.empty(n, C, 2, device=dev); part4 = torch.empty(n * C, 4, device=dev)
    pcoef = torch.empty(n * C, V, device=dev); coeff = torch.randn(V, device=dev)
    xbar = torch.empty(n, C, V, device=dev); dxbar = torch.randn(n, C, V, device=dev)
    S = []
    for _ in range(nsets):
        z = torch.randn(n, C, T, V, device=dev)
        S.append(NS(z=z, zaug=torch.randn(n, C, T, device=dev), x2=torch.randn_like(z), g=torch.randn_like(z),
                    h=torch.empty(n, C, T, V + 1, device=dev), dh=torch.randn(n, C, T, V + 1, device=dev),
                    o=torch.randn(n, C, T, V + 1, device=dev), do=torch.empty(n, C, T, V + 1, device=dev),
                    f=torch.empty_like(z), dz=torch.empty_like(z), dz2=torch.empty_like(z),
                    dzaug=torch.empty(n, C, T, device=dev), out=torch.empty_like(z)))
    rows = []
    def add(name, mb, fn):
        us = timeit(fn)
        rows.append(f'{name:18s} {us:7.1f} us {mb / us * 1e3:7.0f} GB/s')
    na = C - C // 6
    add('branch_act_fwd', pl + pl1, lambda i: lib.dsgcn_branch_act_fwd(P(S[i].z), P(S[i].zaug), P(sc), P(sh), na, P(S[i].h), n, C, T, V, st))
    add('branch_act_bwd', 2 * pl + pl1, lambda i: lib.dsgcn_branch_act_bwd(P(S[i].z), P(S[i].zaug), P(sc), P(sh), na, P(S[i].dh), P(S[i].dz), P(S[i].dzaug), P(part2), n, C, T, V, st))
    add('tms_combine_fwd', pl + pl1, lambda i: lib.dsgcn_tms_combine_fwd(P(S[i].o), P(coeff), P(S[i].f), P(part2), n, C, T, V, st))
    add('tms_combine_bwd', pl + 2 * pl1, lambda i: lib.dsgcn_tms_combine_bwd(P(S[i].o), P(coeff), P(S[i].g), P(A0), P(B0), P(S[i].do), P(pcoef), n, C, T, V, st))
    add('fuse_out_fwd', 3 * pl, lambda i: lib.dsgcn_fuse_out_fwd(P(S[i].z), P(sc), P(sh), P(S[i].x2), P(sc), P(sh), 1, P(S[i].out), P(xbar), n, C, T, V, V, st))
    add('fuse_out_bwd', 5 * pl, lambda i: lib.dsgcn_fuse_out_bwd(P(S[i].z), P(sc), P(sh), P(S[i].x2), P(sc), P(sh), 1, P(S[i].g), P(dxbar), P(S[i].dz), P(S[i].dz2), P(part4), n, C, T, V, V, st))
    add('fuse_out_bwd(id)', 5 * pl, lambda i: lib.dsgcn_fuse_out_bwd(P(S[i].z), P(sc), P(sh), P(S[i].x2), None, None, 1, P(S[i].g), P(dxbar), P(S[i].dz), P(S[i].dz2), P(part4), n, C, T, V, V, st))
    add('dz_eff_aug', 3 * pl, lambda i: lib.dsgcn_dz_eff_aug(P(S[i].g), P(S[i].z), P(S[i].dzaug), P(S[i].zaug), P(A0), P(B0), P(S[i].dz), n, C, T, V, st))
    add('add3', 4 * pl, lambda i: lib.dsgcn_add3(P(S[i].z), P(S[i].x2), P(S[i].g), P(S[i].dz), S[i].z.numel(), st))
    add('copy (torch)', 2 * pl, lambda i: S[i].dz.copy_(S[i].z))
    print(f'--- C={C} T={T}: plane set {pl:.1f} MB, {nsets} rotating operand sets')
    print('\n'.join(rows))
    del S
