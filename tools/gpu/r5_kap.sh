#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_kap; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "aggregate_sum or ctr or aagcn or gram or units_vs" > $O/test_k.log 2>&1; tail -3 $O/test_k.log
timeout 1200 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "ctrgcn or aagcn or stgcn" > $O/test_m.log 2>&1; tail -3 $O/test_m.log
for i in 1 2; do
DSGCN_LAB_LIB=$R/ds-gcn_amd/lib/libdsgcn_lab_old.so timeout 300 python tools/step_ab.py "" --rounds 1 --kind ctrgcn 2>&1 | grep -v amdgpu | sed 's/^/old ctrgcn /'
timeout 300 python tools/step_ab.py "" --rounds 1 --kind ctrgcn 2>&1 | grep -v amdgpu | sed 's/^/new ctrgcn /'
done | tee $O/step_ab.txt
