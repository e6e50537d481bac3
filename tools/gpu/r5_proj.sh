#!/bin/bash
# the dynamic adjacency's projection convs (planes of 32 padded joints): tiny-tile K-C against the GEMM form (lab key 12 = its minimum plane)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_proj; mkdir -p $O; cd $R
S='p1,64,72,1,0;p2,128,144,1,0;p3,256,288,1,0;p0,3,72,1,0'
KC_V=32 KC_SHAPES="$S" timeout 300 python tools/kc_bench.py "" "12=32" "12=32,11=32" 2>&1 | grep -v amdgpu | cut -c1-110 | tee $O/kc.txt
