"""N eager training steps of the bench model (target program for rocprofv3 runs that should see the step only):
    python tools/step_once.py [steps] [phase]     phase: all | fwd (no backward) """
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, dsgcn_amd
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
phase = sys.argv[2] if len(sys.argv) > 2 else 'all'
dev = torch.device('cuda')
model = bench.build_model().to(dev).train()
flat = dsgcn_amd.FlatParams(model, gather=True)
opt = dsgcn_amd.FlatSGD(flat)
g = torch.Generator().manual_seed(0)
batch = dict(keypoint=torch.randn(64, 1, 2, 64, 25, 3, generator=g).to(dev), label=torch.randint(0, 60, (64, 1), generator=g).to(dev))
for _ in range(steps):
    opt.zero_grad()
    out = model.train_step(batch, None, sync_log_vars=False)
    if phase == 'all':
        out['loss'].backward()
        flat.collect_grads()
        opt.step()
torch.cuda.synchronize()
