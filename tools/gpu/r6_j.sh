#!/bin/bash
# round 6: CTR-GCN's three conv4's per unit as one launch each way (k_pw4 grouped over blockIdx.y)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_j; mkdir -p $O; cd $R
PREV=$R/ds-gcn_amd/lib/libdsgcn_lab_prev.so
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "ctr or pwconv" > $O/test_k.log 2>&1; tail -3 $O/test_k.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "ctrgcn or reduced or deferred" > $O/test_m.log 2>&1; tail -3 $O/test_m.log
timeout 600 python tools/step_ab.py "" py:GROUP_CONVS=0 --kind ctrgcn --rounds 3 2>&1 | grep -v amdgpu | sed "s/^/ctrgcn /" | tee $O/step_ab.txt
for i in 1 2; do
DSGCN_LAB_LIB=$PREV timeout 300 python tools/step_ab.py "" --rounds 2 2>&1 | grep -v amdgpu | sed 's/^/ds prev /'
timeout 300 python tools/step_ab.py "" --rounds 2 2>&1 | grep -v amdgpu | sed 's/^/ds new  /'
done | tee -a $O/step_ab.txt
