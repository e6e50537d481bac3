"""Inference throughput (eval-mode BN folded into the consumers' load prologue, no_grad), 64 clips:
    python tools/bench_eval.py [ds|ctrgcn|stgcn|stgcnpp] [clips]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
import dsgcn_amd as D
import bench
from bench import other_cfg
kind = sys.argv[1] if len(sys.argv) > 1 else 'ds'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
np.random.seed(0); torch.manual_seed(0)
m = bench.build_model() if kind == 'ds' else D.build_model(other_cfg(kind))
m = m.cuda().eval()
x = torch.randn(N, 1, 2, 64, 25, 3).cuda()
def step():
    with torch.no_grad():
        return m.cls_head(m.extract_feat(x[:, 0]))
for _ in range(3): step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(20): g.replay()
torch.cuda.synchronize()
dt = (time.time() - t0) / 20
print(f'{kind} eval forward: {dt*1e3:.2f} ms per {N} clips = {N/dt:.0f} clips/s (hipGraph replay)')
