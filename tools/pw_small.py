"""Timing of the tiny K-C launches (projections on the time mean): python tools/pw_small.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsgcn_amd import kernels as K
dev = 'cuda'
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (n, Ci, T, V, Co) in [(128, 64, 1, 25, 72), (128, 128, 1, 25, 144), (128, 256, 1, 25, 288), (128, 3, 1, 25, 72),
                          (128, 8, 25, 25, 64), (128, 32, 25, 25, 256), (128, 64, 64, 25, 64)]:
    x = torch.randn(n, Ci, T, V, device=dev, requires_grad=True)
    w = torch.randn(Co, Ci, device=dev, requires_grad=True); b = torch.randn(Co, device=dev, requires_grad=True)
    f = lambda: K.pwconv(x, None, None, None, False, w, b, 1, False)[0]
    z = f(); g = torch.randn_like(z)
    def fb():
        z = f(); z.backward(g)
    print(f'n={n} Ci={Ci} T={T} V={V} Co={Co}: fwd {timeit(f):7.1f} us   fwd+bwd {timeit(fb):7.1f} us', flush=True)
