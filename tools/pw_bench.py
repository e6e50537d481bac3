"""Timing of the K-C (1x1 conv) kernels on the DS-STGCN layer shapes (n=128)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsgcn_amd import kernels as K, native
dev = 'cuda'
n, V = 128, 25
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
shapes = [  # name, Ci, Co, T, stride, aug, res
    ('pre0', 3, 24, 64, 1, 0, 0), ('pre1', 64, 24, 64, 1, 0, 0), ('post1', 24, 64, 64, 1, 0, 0),
    ('branch1', 64, 64, 64, 1, 1, 1), ('transf1', 64, 64, 64, 1, 0, 0),
    ('pre5', 128, 48, 32, 1, 0, 0), ('post5', 48, 128, 32, 1, 0, 0), ('branch5', 128, 128, 32, 1, 1, 1),
    ('pre8', 256, 96, 16, 1, 0, 0), ('post8', 96, 256, 16, 1, 0, 0), ('branch8', 256, 256, 16, 1, 1, 1),
    ('resid4', 64, 128, 64, 2, 0, 0)]
roll = int(sys.argv[1]) if len(sys.argv) > 1 else 1
native.lib().dsgcn_pwconv_tuning(2, roll)
print('roll', roll)
print(f'{"name":9s} {"fwd us":>8s} {"GB/s":>7s} {"TF/s":>6s} | {"f+b us":>8s} {"TF/s":>6s}')
for name, Ci, Co, T, stride, aug, res in shapes:
    x1 = torch.randn(n, Ci, T, V, device=dev, requires_grad=True)
    a1 = (torch.rand(Ci, device=dev) + .5, torch.randn(Ci, device=dev) * .1)
    x2 = torch.randn(n, Ci, T, V, device=dev, requires_grad=True) if res else None
    w = torch.randn(Co, Ci, 1, 1, device=dev, requires_grad=True) * Ci ** -.5
    w = w.detach().requires_grad_()
    b = torch.zeros(Co, device=dev, requires_grad=True)
    g = torch.ones(Co, device=dev, requires_grad=True); be = torch.zeros(Co, device=dev, requires_grad=True)
    Tout = (T + stride - 1) // stride
    def fwd():
        with torch.no_grad():
            return K.pwconv(x1, a1, x2, None, True, w, b, stride, bool(aug), g, be, 1e-5, Co, True)
    gz = torch.randn(n, Co, Tout, V, device=dev)
    def fb():
        z, zaug, sc, sh, _, _ = K.pwconv(x1, a1, x2, None, True, w, b, stride, bool(aug), g, be, 1e-5, Co, True)
        loss = (z * gz).sum() + sc.sum() + sh.sum()
        loss.backward()
    tf = timeit(fwd)
    tb = timeit(fb)
    flops = 2 * Ci * Co * n * Tout * V
    byts = 4 * n * V * (Ci * T * (2 if res else 1) + Co * Tout)
    print(f'{name:9s} {tf:8.1f} {byts/tf/1e3:7.0f} {flops/tf/1e6:6.1f} | {tb:8.1f} {3*flops/tb/1e6:6.1f}', flush=True)
