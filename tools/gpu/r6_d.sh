#!/bin/bash
# round 6: K-split form of the tiny-plane K-C launches (projections): parity + A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_d; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "dynadj or pwconv or ctr_topology" > $O/test_k.log 2>&1; tail -5 $O/test_k.log
for i in 1 2; do
timeout 600 python tools/step_ab.py "" 19=0 --rounds 2 2>&1 | grep -v amdgpu
done | tee $O/step_ab.txt
timeout 300 python tools/step_ab.py "" 19=0 --kind ctrgcn --rounds 2 2>&1 | grep -v amdgpu | sed "s/^/ctrgcn /" | tee $O/step_ab_ctr.txt
