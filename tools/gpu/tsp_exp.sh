mkdir -p gpurun_out/prof
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "test_temporal_ms and split" -x 2>&1 | tail -2
for e in ${TS_EXPS:-0 1 8 16 32 64 128 256}; do echo "== exp $e"; export TS_EXP=$e; bash tools/gpu/prof_cmd.sh tsp$e 40 21 -- python3 $GRAFT_REPO_ROOT/tools/tsp_bench.py 128 20 2>&1 | grep "k_tsp"; done
