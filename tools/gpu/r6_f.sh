#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_f; mkdir -p $O; cd $R
timeout 600 python tools/step_ab.py "" 15=1 16=512 --kind ctrgcn --rounds 3 2>&1 | grep -v amdgpu | sed "s/^/ctrgcn /" | tee $O/step_ab.txt
