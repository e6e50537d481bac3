#!/bin/bash
# re-record the frozen gradient-ratio marks with the kernels of this tree (twice: the second run must reproduce the first)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_marks; mkdir -p $O; cd $R
for i in 1 2; do
rm -f $O/marks_$i.json
DSGCN_RECORD_GRAD_RATIOS=$O/marks_$i.json timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "full_width_gradients" > $O/test_$i.log 2>&1; tail -1 $O/test_$i.log
done
cmp $O/marks_1.json $O/marks_2.json && echo "marks reproduce"
cat $O/marks_1.json
