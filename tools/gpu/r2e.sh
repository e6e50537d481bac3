#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2e; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -q -m gpu -k "reference_fixture or intermediates or config5 or full_model_vs_oracle or packing" > $O/test.log 2>&1; tail -40 $O/test.log
