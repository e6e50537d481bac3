"""K-A launch geometry judged INSIDE the replayed step (round 6): the lab library stands in for the product library, the
K-A knobs of dsgcn_set_tuning are set from the command line, the bench step is captured and replayed; run it under
rocprofv3 --kernel-trace --stats and read the K-A rows of the stats (tools/gpu/r6_ka_instep.sh does that per variant).
    python tools/ka_instep.py [key=value,...] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, dsgcn_amd
from dsgcn_amd import native

lab = native.lab_lib()
native._lib = lab
variant = sys.argv[1] if len(sys.argv) > 1 else ''
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for kv in [x for x in variant.split(',') if x]:
    k, v = kv.split('=')
    assert lab.dsgcn_set_tuning(int(k), int(v)) == 0, kv
dev = torch.device('cuda')
model = bench.build_model().to(dev).train()
eng = dsgcn_amd.TrainEngine(model, lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True, use_graph=True, warmup_eager=2)
g = torch.Generator().manual_seed(1234)
kp = torch.randn(64, 1, 2, 64, 25, 3, generator=g).to(dev)
lb = torch.randint(0, 60, (64, 1), generator=g).to(dev)
for _ in range(4 + steps):
    eng.step(kp, lb)
torch.cuda.synchronize()
assert eng.graphed(kp, lb), eng.capture_error
