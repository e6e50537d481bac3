#!/bin/bash
# bisect: which of the two batched CTR launches breaks the graph capture of tools/bench_other.py
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_p; mkdir -p $O; cd $R
for c in "0 0" "1 0" "0 1" "1 1"; do
  set -- $c
  echo "== prep=$1 fin=$2"
  DSGCN_CTR_PREP_BATCH=$1 DSGCN_CTR_FIN_BATCH=$2 timeout 300 python tools/bench_other.py ctrgcn 8 5 > $O/run_$1$2.log 2>&1
  echo "rc=$?"; grep -v "^    @" $O/run_$1$2.log | tail -4 | cut -c1-300
done
echo "== engine"
timeout 600 python - <<'PY' 2>&1 | grep -v "^    @" | tail -5
import sys; sys.path.insert(0, '.')
import torch, numpy as np, dsgcn_amd as D
from bench import other_cfg
m = D.build_model(other_cfg('ctrgcn')).cuda().train()
eng = D.TrainEngine(m, lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True, use_graph=True)
x = torch.randn(8, 1, 2, 64, 25, 3).cuda(); y = torch.randint(0, 60, (8, 1)).cuda()
for i in range(8):
    out = eng.step(x, y)
torch.cuda.synchronize(); print('engine ok', {k: float(v) for k, v in out.items()})
PY
