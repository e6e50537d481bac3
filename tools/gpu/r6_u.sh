#!/bin/bash
# k_tcw's wave mapping for 64-row co tiles: the ST-GCN step with / without (lab knob tc:1), same box, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_u; mkdir -p $O; cd $R
timeout 900 python tools/step_ab.py 'tc:1=0' '' --kind stgcn --steps 20 --rounds 3 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|run_backward" | tee $O/ab.txt
