import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench
from oracle import dsgcn_oracle as O
thr = int(sys.argv[1]); batch = int(sys.argv[2])
torch.set_num_threads(thr)
model = bench.build_model()
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
leaves = {k: v.requires_grad_() for k, v in sd.items() if v.dtype.is_floating_point and 'running' not in k}
sd.update(leaves)
gc = O.graph_constants('nturgb+d'); plan = O.dgstgcn_plan()
x = torch.randn(batch,1,2,64,25,3); y = torch.randint(0,60,(batch,1))
for i in range(3):
    t=time.perf_counter()
    _, loss = O.recognizer_forward_train(x, y, sd, gc['node_type'], gc['edge_type'], plan)
    t1=time.perf_counter()
    loss.backward()
    print(thr, batch, 'fwd %.2f bwd %.2f  clips/s %.1f'%(t1-t, time.perf_counter()-t1, batch/(time.perf_counter()-t)), flush=True)
