#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_i; mkdir -p $O; cd $R
timeout 900 python tools/step_ab.py "" 22=1 22=2 22=3 --rounds 3 2>&1 | grep -v amdgpu | tee $O/step_ab.txt
