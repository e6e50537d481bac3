#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_k; mkdir -p $O; cd $R
PREV=$R/ds-gcn_amd/lib/libdsgcn_lab_prev.so
for i in 1 2 3; do
DSGCN_LAB_LIB=$PREV timeout 300 python tools/step_ab.py "" --rounds 2 2>&1 | grep -v amdgpu | grep ms/step | sed 's/^/ds prev /'
timeout 300 python tools/step_ab.py "" --rounds 2 2>&1 | grep -v amdgpu | grep ms/step | sed 's/^/ds new  /'
done | tee $O/step_ab.txt
