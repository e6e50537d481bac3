#!/bin/bash
# end of round 5 (second half): rocprof stats of configs 1 and 4 and the kernel sequences of the three steps
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_final2; mkdir -p $O; cd $R
for k in stgcn ctrgcn; do
  bash tools/gpu/prof_cmd.sh $k 14 17 -- python3 $R/tools/bench_other.py $k 64 10 > $O/prof_$k.txt 2>&1
  cp $R/gpurun_out/prof/${k}_kernel_stats.csv $O/${k}_kernel_stats.csv
  tail -3 $O/prof_$k.txt
  bash tools/gpu/r5_seq_other.sh $k > /dev/null 2>&1; cp $R/gpurun_out/r5_seq_$k/sequence.txt $O/step_sequence_$k.txt
done
bash tools/gpu/r5_seq.sh > /dev/null 2>&1; cp $R/gpurun_out/r5_seq/sequence.txt $O/step_sequence.txt
bash tools/gpu/r5_gaps.sh > $O/trace_gaps.txt 2>&1; tail -3 $O/trace_gaps.txt
wc -l $O/step_sequence*.txt
