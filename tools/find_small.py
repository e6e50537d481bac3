"""Where do the tiny elementwise launches of a step come from?  python tools/find_small.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, dsgcn_amd
from torch.profiler import profile, ProfilerActivity
m = bench.build_model().cuda().train()
flat = dsgcn_amd.FlatParams(m, gather=True)
g = torch.Generator().manual_seed(0)
x = torch.randn(16, 1, 2, 64, 25, 3, generator=g).cuda(); y = torch.randint(0, 60, (16, 1), generator=g).cuda()
def step():
    flat.zero_grad()
    out = m.train_step(dict(keypoint=x, label=y), None, sync_log_vars=False)
    out['loss'].backward()
    flat.collect_grads()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::zero_', 'aten::clone', 'aten::cat', 'aten::fill_', 'aten::copy_', 'aten::add', 'aten::add_', 'aten::mul', 'aten::sum'):
        chain = []
        p = ev.cpu_parent
        while p is not None and len(chain) < 3:
            chain.append(p.name)
            p = p.cpu_parent
        cnt[(ev.name, tuple(chain))] += 1
for (k, st), v in cnt.most_common(40):
    print(v, k, ' <- '.join(st))
