#!/bin/bash
# everything a round's profiles/rNN/ directory holds beyond profile_set.sh: the other configurations (step time + kernel
# stats for three of them), the input pipeline, K-B / K-C micro-benchmarks, and the numbers the parity tests print.
#   gpurun --timeout 2400 -- 'bash tools/gpu/round_set.sh'      -> gpurun_out/round_set/
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/round_set; mkdir -p $O
cd $R
for k in stgcn stgcnpp ctrgcn ctrgcn_shipped stgcn_shipped aagcn dggcn ds120 ds_k400; do
  timeout 300 python tools/bench_other.py $k 2>&1 | tail -1
done > $O/other_configs.txt
cat $O/other_configs.txt
timeout 300 python tools/pipeline_bench.py > $O/pipeline.txt 2>&1; tail -8 $O/pipeline.txt
timeout 600 python tools/train_bench.py > $O/train_bench.txt 2>&1; tail -5 $O/train_bench.txt
KC_PHASES=1 timeout 400 python tools/kc3.py 2 wgrad 2>&1 | grep -v amdgpu.ids > $O/kc3.txt; tail -30 $O/kc3.txt
timeout 600 python tools/step_ab.py "" 14=1,15=0 --rounds 2 2>&1 | grep -v amdgpu.ids > $O/step_ab.txt; cat $O/step_ab.txt
KB_N=128 timeout 200 python tools/kb_bench.py 2>&1 | grep mid > $O/kb_bench.txt; cat $O/kb_bench.txt
timeout 300 python tools/kc_bench.py 2>&1 | grep -v amdgpu.ids > $O/kc_bench.txt; tail -16 $O/kc_bench.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_train_loop.py -q -s -k "bf16_split or full_width_gradients or rccl or trajectory or torch_distributed or step_lr" 2>&1 | grep -v "^$" | tail -60 > $O/parity_numbers.txt
tail -40 $O/parity_numbers.txt
for k in ds_k400 ctrgcn stgcn; do
  bash tools/gpu/prof_cmd.sh $k 12 10 -- python3 $R/tools/bench_other.py $k $([ $k = ds_k400 ] && echo 32 || echo 64) 10 > $O/prof_$k.txt 2>&1
  cp $R/gpurun_out/prof/${k}_kernel_stats.csv $O/${k}_kernel_stats.csv
  tail -14 $O/prof_$k.txt
done
