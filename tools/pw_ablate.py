import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsgcn_amd import native, kernels as K
lib = native.lib(); dev = 'cuda'; st = torch.cuda.current_stream().cuda_stream
n, V = 128, 25
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (Ci, Co, T, res) in [(64, 64, 64, 0), (64, 64, 64, 1), (128, 128, 32, 0), (256, 256, 16, 0)]:
    x1 = torch.randn(n, Ci, T, V, device=dev); x2 = torch.randn(n, Ci, T, V, device=dev) if res else None
    s1 = torch.rand(Ci, device=dev) + .5; h1 = torch.randn(Ci, device=dev) * .1
    w = torch.randn(Co, Ci, device=dev) * Ci ** -.5; b = torch.zeros(Co, device=dev)
    z = torch.empty(n, Co, T, V, device=dev)
    part = torch.empty(lib.dsgcn_pwconv_partial_rows(n, Co, T, V, 1, 0) * 0 + 4096, Co, 2, device=dev)
    def f():
        rc = lib.dsgcn_pwconv_fwd(x1.data_ptr(), s1.data_ptr(), h1.data_ptr(), x2.data_ptr() if res else None, None, None, 1,
                                  w.data_ptr(), b.data_ptr(), z.data_ptr(), None, part.data_ptr(), n, Ci, Co, T, V, 1, 0, 1, st)
        assert rc == 0
    for mt in (2, 1):
        lib.dsgcn_pwconv_tuning(1, mt)
        line = f'Ci={Ci} Co={Co} T={T} res={res} maxMT={mt}: '
        for ab, nm in ((0, 'full'), (4, '-store'), (8, '-stats'), (12, '-store-stats')):
            lib.dsgcn_pwconv_tuning(0, ab)
            line += f'{nm} {timeit(f):6.1f} | '
        lib.dsgcn_pwconv_tuning(0, 0)
        print(line, flush=True)
    lib.dsgcn_pwconv_tuning(1, 2)
