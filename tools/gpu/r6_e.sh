#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_e; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "pwconv or ctr_topology" > $O/test_k.log 2>&1; tail -3 $O/test_k.log
for k in ctrgcn ds stgcn; do
timeout 400 python tools/step_ab.py "" 20=0 --kind $k --rounds 3 2>&1 | grep -v amdgpu | sed "s/^/$k /"
done | tee $O/step_ab.txt
