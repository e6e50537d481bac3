#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_seq; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/raw -o p --output-format csv -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-roofline --no-other-configs > $O/run.log 2>&1
f=$(find $O/raw -name 'p_kernel_trace.csv' | head -1)
python3 $R/tools/step_sequence.py "$f" $O/sequence.txt; tail -24 $O/sequence.txt
rm -rf $O/raw
