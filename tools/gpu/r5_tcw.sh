#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_tcw; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "tconv_gemm or units_vs_reference or aagcn" > $O/test_k.log 2>&1; tail -3 $O/test_k.log
for i in 1 2; do
TC_LAB=1 DSGCN_LAB_LIB=$R/ds-gcn_amd/lib/libdsgcn_lab_old.so timeout 300 python tools/tcg_bench.py 2>&1 | grep -v amdgpu | grep "^s" | cut -c1-110 | sed 's/^/old /'
TC_LAB=1 timeout 300 python tools/tcg_bench.py 2>&1 | grep -v amdgpu | grep "^s" | cut -c1-110 | sed 's/^/new /'
done | tee $O/tcg.txt
for i in 1 2; do
DSGCN_LAB_LIB=$R/ds-gcn_amd/lib/libdsgcn_lab_old.so timeout 300 python tools/step_ab.py "" --rounds 1 --kind stgcn 2>&1 | grep -v amdgpu | sed 's/^/old stgcn /'
timeout 300 python tools/step_ab.py "" --rounds 1 --kind stgcn 2>&1 | grep -v amdgpu | sed 's/^/new stgcn /'
done | tee $O/step_ab.txt
