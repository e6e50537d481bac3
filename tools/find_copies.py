"""Where do the per-step D2D blits / column sums come from?  torch.profiler over one eager bench step, aten::copy_ /
aten::contiguous / aten::clone / aten::cat grouped by the calling Python line."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, dsgcn_amd
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda')
model = bench.build_model().to(dev).train()
flat = dsgcn_amd.FlatParams(model, gather=True)
opt = dsgcn_amd.FlatSGD(flat)
g = torch.Generator().manual_seed(0)
batch = dict(keypoint=torch.randn(64, 1, 2, 64, 25, 3, generator=g).to(dev), label=torch.randint(0, 60, (64, 1), generator=g).to(dev))
def step():
    opt.zero_grad(); out = model.train_step(batch, None, sync_log_vars=False); out['loss'].backward(); flat.collect_grads(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::contiguous', 'aten::clone', 'aten::cat', 'aten::zeros', 'aten::zero_', 'aten::fill_', 'aten::add', 'aten::mul', 'aten::pad', 'aten::constant_pad_nd'):
        frames = [f for f in (ev.stack or []) if 'ds-gcn_amd' in f or 'bench.py' in f or 'autograd' in f]
        key = (ev.name, frames[0].split('/')[-1] if frames else 'no-stack')
        cnt[key] += 1
for (name, where), c in cnt.most_common(40):
    print(f'{c:5d}  {name:24s} {where}')
