#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_h; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -s -k "full_width_gradients" 2>&1 | grep "ratio\|passed\|failed" | tee $O/marks.log
