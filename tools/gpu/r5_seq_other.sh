#!/bin/bash
# kernel sequence of one replayed step of another configuration:  bash tools/gpu/r5_seq_other.sh ctrgcn|stgcn
R=${GRAFT_REPO_ROOT:-/root/repo}; K=${1:-ctrgcn}; O=$R/gpurun_out/r5_seq_$K; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/raw -o p --output-format csv -- python3 $R/tools/bench_other.py $K 64 6 > $O/run.log 2>&1
f=$(find $O/raw -name 'p_kernel_trace.csv' | head -1)
python3 $R/tools/step_sequence.py "$f" $O/sequence.txt; tail -24 $O/sequence.txt
rm -rf $O/raw
