#!/bin/bash
# round 6: fused dropout parity + everything else
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_c; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "fuse_out" > $O/test_k.log 2>&1; tail -5 $O/test_k.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "dropout or reduced or full_other" > $O/test_m.log 2>&1; tail -5 $O/test_m.log
timeout 2400 python -m pytest tests -q -m gpu > $O/test_all.log 2>&1; tail -5 $O/test_all.log
