#!/bin/bash
# round 5, first GPU call: the tests touched so far + the K400 input pipeline line + a bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_first; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_pipeline.py tests/test_train_loop.py tests/test_data_parallel.py -q -m gpu -x > $O/test_a.log 2>&1; tail -3 $O/test_a.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "fuse_out" > $O/test_b.log 2>&1; tail -3 $O/test_b.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x > $O/test_c.log 2>&1; tail -3 $O/test_c.log
timeout 300 python tools/pipeline_bench.py 2048 32 k400 > $O/pipeline_k400.txt 2>&1; tail -8 $O/pipeline_k400.txt
timeout 300 python tools/pipeline_bench.py > $O/pipeline.txt 2>&1; tail -7 $O/pipeline.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-600 $O/bench.json
