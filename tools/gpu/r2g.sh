#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in 0 1; do
  DSGCN_WGRAD_SIDE=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-200
done
