"""Census of the deferred parameter-gradient column sums of one training step: which partial-row buffers the end-of-backward
dsgcn_colsum_multi launch reads (rows x columns, MB), largest first.
    python tools/param_sum_census.py [--kind ds|stgcn|ctrgcn]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, dsgcn_amd
from dsgcn_amd import kernels
from dsgcn_amd.engine import TrainEngine

kind = sys.argv[sys.argv.index('--kind') + 1] if '--kind' in sys.argv else 'ds'
dev = torch.device('cuda')
if kind == 'ds':
    model = bench.build_model().to(dev).train()
else:
    import numpy as np
    np.random.seed(0); torch.manual_seed(0)
    model = dsgcn_amd.build_model(bench.other_cfg(kind)).to(dev).train()
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
g = torch.Generator().manual_seed(1234)
kp = torch.randn(64, 1, 2, 64, 25, 3, generator=g).to(dev)
lb = torch.randint(0, 60, (64, 1), generator=g).to(dev)
eng = TrainEngine(model, use_graph=False)
seen = []
orig = kernels._flush_jobs


def spy(jobs):
    seen.append([(R, C) for _, R, C, _ in jobs])
    return orig(jobs)


kernels._flush_jobs = spy
eng.step(kp, lb)
torch.cuda.synchronize()
seen.clear()
eng.step(kp, lb)
torch.cuda.synchronize()
for level, jobs in enumerate(seen):
    tot = sum(R * C for R, C in jobs) * 4 / 2 ** 20
    print(f'launch {level}: {len(jobs)} jobs, {tot:.1f} MiB')
    agg = {}
    for R, C in jobs:
        agg[(R, C)] = agg.get((R, C), 0) + 1
    for (R, C), cnt in sorted(agg.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[1])[:16]:
        print(f'   {cnt:3d} x  {R:6d} rows x {C:7d} cols = {cnt * R * C * 4 / 2 ** 20:8.1f} MiB')
