#!/bin/bash
# conv4 of the CTR-GCN refinement at R, R+1, R+2 input channels (planes of V x V = 625 positions), without / with input affine
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_c4; mkdir -p $O; cd $R
S='a8,8,64,25,0;a9,9,64,25,0;a10,10,64,25,0;a10f,10,64,25,1;b16,16,128,25,0;b17,17,128,25,0;b18,18,128,25,0;b18f,18,128,25,1;c32,32,256,25,0;c33,33,256,25,0;c34,34,256,25,0;c34f,34,256,25,1'
KC_SHAPES="$S" timeout 300 python tools/kc_bench.py 2>&1 | grep -v amdgpu | cut -c1-110 | tee $O/kc.txt
