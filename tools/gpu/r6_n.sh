#!/bin/bash
# round 6: stride-2 fused temporal data gradient on parity tiles
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_n; mkdir -p $O; cd $R
PREV=$R/ds-gcn_amd/lib/libdsgcn_lab_prev.so
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "temporal" > $O/test_k.log 2>&1; tail -3 $O/test_k.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "ctrgcn or reduced" > $O/test_m.log 2>&1; tail -3 $O/test_m.log
for i in 1 2; do
DSGCN_LAB_LIB=$PREV timeout 300 python tools/step_ab.py "" --kind ctrgcn --rounds 2 2>&1 | grep -v amdgpu | grep ms/step | sed 's/^/ctrgcn prev /'
timeout 300 python tools/step_ab.py "" --kind ctrgcn --rounds 2 2>&1 | grep -v amdgpu | grep ms/step | sed 's/^/ctrgcn new  /'
done | tee $O/step_ab.txt
