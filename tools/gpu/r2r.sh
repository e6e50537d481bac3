#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2r; mkdir -p $O; cd $R
for k in ds120 ds_k400 stgcn ctrgcn ctrgcn_shipped stgcnpp; do timeout 300 python tools/bench_other.py $k 2>&1 | tail -1; done | tee $O/other.log
