#!/bin/bash
# kernel sequence of one replayed step: bash tools/gpu/r6_seq.sh [ds|stgcn|ctrgcn]
R=${GRAFT_REPO_ROOT:-/root/repo}; K=${1:-ds}; O=$R/gpurun_out/r6_seq_$K; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
if [ "$K" = ds ]; then
timeout 600 rocprofv3 --kernel-trace -d $O/raw -o p --output-format csv -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-roofline --no-other-configs > $O/run.log 2>&1
else
timeout 600 rocprofv3 --kernel-trace -d $O/raw -o p --output-format csv -- python3 $R/tools/bench_other.py $K 64 6 > $O/run.log 2>&1
fi
f=$(find $O/raw -name 'p_kernel_trace.csv' | head -1)
python3 $R/tools/step_sequence.py "$f" $O/sequence.txt; tail -12 $O/sequence.txt; grep -c "us " $O/sequence.txt
rm -rf $O/raw
