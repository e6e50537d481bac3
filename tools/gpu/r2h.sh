#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2h; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -q -m gpu -k "units_vs_reference or shipped or eval_mode or full_width or other_backbones" > $O/test.log 2>&1; tail -25 $O/test.log
