#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_bwd64; mkdir -p $O; cd $R
S='pre1,64,24,64,0;post1,24,64,64,1;branch1,64,64,64,2;transf1,64,64,64,1;in0,3,24,64,0'
for i in 1 2; do
KC_SHAPES="$S" DSGCN_LAB_LIB=$R/ds-gcn_amd/lib/libdsgcn_lab_old.so timeout 300 python tools/kc_bench.py 2>&1 | grep -v amdgpu | grep fused | sed 's/^/old /'
KC_SHAPES="$S" timeout 300 python tools/kc_bench.py 2>&1 | grep -v amdgpu | grep fused | sed 's/^/new /'
done | tee $O/kc.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "pwconv" > $O/test_k.log 2>&1; tail -3 $O/test_k.log
