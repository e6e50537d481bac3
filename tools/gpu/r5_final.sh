#!/bin/bash
# end of round 5: the bench line with the final kernels + rocprof stats of configs 1 and 4
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_final; mkdir -p $O; cd $R
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench.err; cut -c1-300 $O/bench_n1.json
for k in stgcn ctrgcn; do
  bash tools/gpu/prof_cmd.sh $k 14 17 -- python3 $R/tools/bench_other.py $k 64 10 > $O/prof_$k.txt 2>&1
  cp $R/gpurun_out/prof/${k}_kernel_stats.csv $O/${k}_kernel_stats.csv
  tail -16 $O/prof_$k.txt
done
