#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2p; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/raw -o p --output-format csv -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/run.log 2>&1
f=$(find $O/raw -name 'p_kernel_trace.csv' | head -1)
python3 $R/tools/trace_gaps.py $f > $O/gaps.log 2>&1
rm -rf $O/raw; cat $O/gaps.log; tail -2 $O/run.log | cut -c1-200
