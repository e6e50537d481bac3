#!/bin/bash
# round 6, first set: epilogue prefetch depths without scratch (A/B against round 5's lab library), weight-gradient split
# targets, the frozen gradient-ratio marks
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_a; mkdir -p $O; cd $R
OLD=$R/ds-gcn_amd/lib/libdsgcn_lab_r5.so
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "pwconv or tconv" > $O/test_k.log 2>&1; tail -3 $O/test_k.log
rm -f $O/grad_ratio_marks.json
DSGCN_RECORD_GRAD_RATIOS=$O/grad_ratio_marks.json timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "full_width_gradients" > $O/test_marks.log 2>&1; tail -2 $O/test_marks.log
cat $O/grad_ratio_marks.json
S='pre5,128,48,32,0;post5,48,128,32,1;branch5,128,128,32,2;transf5,128,128,32,1;pre8,256,96,16,0;post8,96,256,16,1;branch8,256,256,16,2;transf8,256,256,16,1'
for i in 1 2; do
KC_SHAPES="$S" DSGCN_LAB_LIB=$OLD timeout 300 python tools/kc_bench.py 2>&1 | grep -v amdgpu | sed 's/^/old /'
KC_SHAPES="$S" timeout 300 python tools/kc_bench.py 2>&1 | grep -v amdgpu | sed 's/^/new /'
done > $O/kc.txt; grep -c . $O/kc.txt
for i in 1 2; do
DSGCN_LAB_LIB=$OLD timeout 300 python tools/step_ab.py "" --rounds 2 2>&1 | grep -v amdgpu | sed 's/^/old /'
timeout 600 python tools/step_ab.py "" 9=256 9=384 7=256 16=128 --rounds 2 2>&1 | grep -v amdgpu | sed 's/^/new /'
done | tee $O/step_ab.txt
for k in stgcn ctrgcn; do
DSGCN_LAB_LIB=$OLD timeout 300 python tools/step_ab.py "" --kind $k --rounds 2 2>&1 | grep -v amdgpu | sed "s/^/old $k /"
timeout 300 python tools/step_ab.py "" --kind $k --rounds 2 2>&1 | grep -v amdgpu | sed "s/^/new $k /"
done | tee $O/step_ab_other.txt
