#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_bwd64b; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "pwconv" > $O/test_k.log 2>&1; tail -4 $O/test_k.log
S='pre1,64,24,64,0;post1,24,64,64,1;branch1,64,64,64,2;transf1,64,64,64,1;in0,3,24,64,0'
for i in 1 2; do
KC_SHAPES="$S" DSGCN_LAB_LIB=$R/ds-gcn_amd/lib/libdsgcn_lab_old.so timeout 300 python tools/kc_bench.py 2>&1 | grep -v amdgpu | grep fused | cut -c1-20,95- | sed 's/^/old /'
KC_SHAPES="$S" timeout 300 python tools/kc_bench.py 2>&1 | grep -v amdgpu | grep fused | cut -c1-20,95- | sed 's/^/new /'
done | tee $O/kc.txt
for i in 1 2; do
DSGCN_LAB_LIB=$R/ds-gcn_amd/lib/libdsgcn_lab_old.so timeout 300 python tools/step_ab.py "" --rounds 2 2>&1 | grep -v amdgpu | sed 's/^/old /'
timeout 300 python tools/step_ab.py "" --rounds 2 2>&1 | grep -v amdgpu | sed 's/^/new /'
done | tee $O/step_ab.txt
