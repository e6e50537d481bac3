#!/bin/bash
# per-step launches of parameter-only work (kernels._StepBuilt): tests, then ST-GCN / CTR-GCN step A/B and sequences
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_s; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "ctr or tconv or tcg or temporal_conv" > $O/test_k.log 2>&1; tail -3 $O/test_k.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "ctrgcn or stgcn or reduced or engine" > $O/test_m.log 2>&1; tail -3 $O/test_m.log
for i in 1 2 3; do
  DSGCN_TSPLIT_BATCH=0 python tools/bench_other.py stgcn 64 20 2>&1 | grep "ms/step" | sed 's/^/per conv: /'
  python tools/bench_other.py stgcn 64 20 2>&1 | grep "ms/step" | sed 's/^/batched:  /'
done | tee $O/ab.txt
bash tools/gpu/r6_seq.sh stgcn > $O/seq2.log 2>&1; tail -3 $O/seq2.log
