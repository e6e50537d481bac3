"""Which host-side ops issue the device-to-device copies (hipMemcpyAsync -> __amd_rocclr_copyBuffer) of a step?
torch.profiler over one eager step; every hipMemcpy* runtime call is attributed to the innermost enclosing CPU op /
autograd node.      python tools/find_memcpy.py"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
import dsgcn_amd  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

dev = torch.device('cuda')
model = bench.build_model().to(dev).train()
eng = dsgcn_amd.TrainEngine(model, use_graph=False)
g = torch.Generator().manual_seed(0)
kp = torch.randn(64, 1, 2, 64, 25, 3, generator=g).to(dev)
lb = torch.randint(0, 60, (64, 1), generator=g).to(dev)
for _ in range(2):
    eng.step(kp, lb)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    eng.step(kp, lb)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]
copies = [e for e in evs if 'Memcpy' in e.name or 'memcpy' in e.name]
ops = [e for e in evs if not ('Memcpy' in e.name or 'memcpy' in e.name or e.name.startswith('hip'))]
count = collections.Counter()
for c in copies:
    t = c.time_range.start
    encl = [o for o in ops if o.time_range.start <= t <= o.time_range.end]
    encl.sort(key=lambda o: o.time_range.end - o.time_range.start)
    names = [o.name for o in encl[:3]]
    count[(c.name, ' < '.join(names))] += 1
print(len(copies), 'memcpy runtime calls in the step')
for (nm, chain), k in count.most_common(30):
    print(f'{k:5d} {nm:24s} {chain}')
