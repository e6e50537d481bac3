#!/bin/bash
# end-of-round refresh: full validation (tests, smoke, default bench) + the rocprof stats of the two other BASELINE steps
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_r; mkdir -p $O; cd $R
bash tools/gpu/validate.sh
cd /tmp; export TMPDIR=/tmp
S() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats -d $O/raw_$name -o p --output-format csv -- "$@" > $O/$name.log 2>&1
      f=$(find $O/raw_$name -name 'p_kernel_stats.csv' | head -1); cp "$f" $O/$name.csv; rm -rf $O/raw_$name; }
S stgcn_kernel_stats python3 $R/tools/bench_other.py stgcn 64 8
S ctrgcn_kernel_stats python3 $R/tools/bench_other.py ctrgcn 64 8
head -5 $O/ctrgcn_kernel_stats.csv | cut -c1-200
