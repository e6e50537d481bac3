#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_m; mkdir -p $O; cd $R
timeout 600 python tools/step_ab.py "" 20=0 --kind ctrgcn --rounds 4 2>&1 | grep -v amdgpu | sed "s/^/ctrgcn /" | tee $O/step_ab.txt
