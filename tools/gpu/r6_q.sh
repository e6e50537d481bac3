#!/bin/bash
# K-B (projection conv + dynamic adjacency) beside the `pre` conv on a second stream, against the default (K-B behind the
# conv, hosting its BatchNorm finalize): same box, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_q; mkdir -p $O; cd $R
timeout 900 python tools/step_ab.py '' py:OVERLAP=1 --steps 20 --rounds 3 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|run_backward" | tee $O/ab.txt
