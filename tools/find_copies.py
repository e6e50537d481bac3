"""Which host-side ATen ops launch the non-library kernels of a step?  torch.profiler (CPU+CUDA) over one eager step
split into phases; prints per (phase, op) the number of device kernels that are NOT from libdsgcn (k_*)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, dsgcn_amd
from torch.profiler import profile, ProfilerActivity, record_function
dev = torch.device('cuda')
model = bench.build_model().to(dev).train()
flat = dsgcn_amd.FlatParams(model, gather=True)
opt = dsgcn_amd.FlatSGD(flat)
g = torch.Generator().manual_seed(0)
batch = dict(keypoint=torch.randn(64, 1, 2, 64, 25, 3, generator=g).to(dev), label=torch.randint(0, 60, (64, 1), generator=g).to(dev))
def step():
    with record_function('PH_zero'): opt.zero_grad()
    with record_function('PH_fwd'): out = model.train_step(batch, None, sync_log_vars=False)
    with record_function('PH_bwd'): out['loss'].backward()
    with record_function('PH_collect'): flat.collect_grads()
    with record_function('PH_opt'): opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
evs = prof.events()
phases = [(e.name, e.time_range.start, e.time_range.end) for e in evs if e.name.startswith('PH_')]
ops = collections.Counter()
for e in evs:
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels and e.name.startswith('aten::'):
        ks = [k for k in e.kernels if 'k_' not in k.name[:40] or 'at::native' in k.name]
        if not ks:
            continue
        # count only leaf ops (the innermost aten op that owns the kernel)
        if any(c.kernels for c in e.cpu_children if c.name.startswith('aten::')):
            continue
        ph = next((p[0] for p in phases if p[1] <= e.time_range.start <= p[2]), '?')
        shp = ''
        ops[(ph, e.name)] += len(ks)
for (ph, name), c in ops.most_common(40):
    print(f'{c:5d} {ph:12s} {name}')
