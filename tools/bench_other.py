"""Step time (fwd + bwd, hipGraph replay) of the other configurations that run on the same kernels:
    python tools/bench_other.py KIND [clips] [steps]
KIND: stgcn (BASELINE config 1), ctrgcn (config 4, classic), ctrgcn_shipped (configs/ctrgcn/CTRGCN_model.py), stgcnpp, aagcn, dggcn,
ds120 (config 3 per-GPU: DS-STGCN NTU-120), ds_k400 (config 5 per-GPU: DS-STGCN coco V=17 T=100, 400 classes, 32 clips)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dsgcn_amd as D
from bench import ds_cfg, other_cfg

kind = sys.argv[1] if len(sys.argv) > 1 else 'ctrgcn'
T, V, classes, dflt = 64, 25, 60, 64
if kind == 'ds_k400':
    cfg, T, V, classes, dflt = ds_cfg(400, 'coco'), 100, 17, 400, 32
elif kind == 'ds120':
    cfg, classes = ds_cfg(120), 120
else:
    cfg = other_cfg(kind)
N = int(sys.argv[2]) if len(sys.argv) > 2 else dflt
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
eager = os.environ.get('DSGCN_EAGER') == '1'
np.random.seed(0); torch.manual_seed(0)
m = D.build_model(cfg)
gen = torch.Generator().manual_seed(1)
with torch.no_grad():
    for k, p in m.named_parameters():
        if k.endswith(('alpha', 'beta', 'add_coeff')):
            p.copy_(torch.randn(p.shape, generator=gen) * 0.5)
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
m = m.cuda().train()
flat = D.FlatParams(m, gather=True)
x = torch.randn(N, 1, 2, T, V, 3, generator=gen).cuda()
y = torch.randint(0, classes, (N, 1), generator=gen).cuda()
def step():
    flat.zero_grad()
    D.kernels.reset_leaf_uses()                    # the step bracket of TrainEngine._fwd_bwd: the parameter-only launches
    try:                                           # (weight images, CTR operands) are batched only inside it
        out = m.train_step(dict(keypoint=x, label=y), None, sync_log_vars=False)
        with D.kernels.deferred_param_sums():      # parameter-gradient column sums in one launch
            out['loss'].backward()
        flat.collect_grads()
    finally:
        D.kernels.end_step()
for _ in range(3):
    step()
torch.cuda.synchronize()
run = step
if not eager:
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    run = g.replay
for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    run()
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print(f'{kind}: {dt*1e3:.2f} ms/step (fwd+bwd{", eager" if eager else ", hipGraph"}), {N} clips -> {N/dt:.1f} clips/s')
