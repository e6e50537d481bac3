"""Host-side ATen op counts of one eager training step (torch.profiler, CPU events): what, besides the C-ABI launches, the
step asks of PyTorch.      python tools/op_counts.py"""
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
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    eng.step(kp, lb)
    torch.cuda.synchronize()
cnt = collections.Counter()
shapes = collections.defaultdict(collections.Counter)
for e in prof.events():
    if e.name.startswith('aten::'):
        cnt[e.name] += 1
        shapes[e.name][str(e.input_shapes)[:90]] += 1
for k, v in cnt.most_common(30):
    print(f'{v:5d} {k}')
    if k in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::cat', 'aten::_to_copy', 'aten::add_', 'aten::add'):
        for sh, c in shapes[k].most_common(8):
            print(f'         {c:4d} x {sh}')
