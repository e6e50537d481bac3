"""Step time of the other two backbones on the native kernels (BASELINE configs 1 and 4 at 64 clips):
    python tools/bench_other.py ctrgcn|stgcn [clips] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch
import dsgcn_amd as D
from bench import other_cfg

kind = sys.argv[1] if len(sys.argv) > 1 else 'ctrgcn'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
np.random.seed(0); torch.manual_seed(0)
m = D.build_model(other_cfg(kind))
with torch.no_grad():
    for k, p in m.named_parameters():
        if k.endswith('alpha'):
            p.normal_(0, 0.5)
m = m.cuda().train()
g = torch.Generator().manual_seed(1)
x = torch.randn(N, 1, 2, 64, 25, 3, generator=g).cuda()
y = torch.randint(0, 60, (N, 1), generator=g).cuda()
def step():
    for p in m.parameters():
        p.grad = None
    out = m.train_step(dict(keypoint=x, label=y), None, sync_log_vars=False) if 'sync_log_vars' in m.train_step.__code__.co_varnames else m.train_step(dict(keypoint=x, label=y), None)
    out['loss'].backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print(f'{kind}: {dt*1e3:.2f} ms/step, {N/dt:.1f} clips/s')
