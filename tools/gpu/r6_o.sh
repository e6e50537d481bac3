#!/bin/bash
# batched CTR-GCN operands / finishing launch: tests, then CTR-GCN step A/B (switches off = one launch per unit) and sequence
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_o; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "ctr" > $O/test_k.log 2>&1; tail -3 $O/test_k.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "ctrgcn or reduced" > $O/test_m.log 2>&1; tail -3 $O/test_m.log
for i in 1 2 3; do
  DSGCN_CTR_PREP_BATCH=0 DSGCN_CTR_FIN_BATCH=0 python tools/bench_other.py ctrgcn 64 20 2>&1 | grep "ms/step" | sed 's/^/per unit: /'
  python tools/bench_other.py ctrgcn 64 20 2>&1 | grep "ms/step" | sed 's/^/batched:  /'
done | tee $O/ab.txt
bash tools/gpu/r6_seq.sh ctrgcn > $O/seq.log 2>&1; tail -3 $O/seq.log
bash tools/gpu/r6_seq.sh stgcn > $O/seq2.log 2>&1; tail -3 $O/seq2.log
