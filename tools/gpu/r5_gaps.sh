#!/bin/bash
# where is the replayed step idle?  kernel trace of the graph-replayed bench step -> tools/trace_gaps.py
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_gaps; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/raw -o p --output-format csv -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-roofline --no-other-configs > $O/run.log 2>&1
f=$(find $O/raw -name 'p_kernel_trace.csv' | head -1)
python3 $R/tools/trace_gaps.py "$f" > $O/gaps.txt 2>&1; cat $O/gaps.txt
rm -rf $O/raw
