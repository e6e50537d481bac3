#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2i; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "aggregate" 2>&1 | tail -2
for k in ds_k400 ds120 stgcn ctrgcn ctrgcn_shipped stgcnpp; do timeout 300 python tools/bench_other.py $k 2>&1 | tail -1; done | tee $O/other.log
cd /tmp; export TMPDIR=/tmp
DSGCN_EAGER=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/raw -o p --output-format csv -- python3 $R/tools/bench_other.py ds_k400 32 5 > $O/prof.log 2>&1
cp $O/raw/*/p_kernel_stats.csv $O/k400_kernel_stats.csv 2>/dev/null || cp $O/raw/p_kernel_stats.csv $O/k400_kernel_stats.csv
rm -rf $O/raw
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open('$O/k400_kernel_stats.csv')))
for r in rows[:14]:
    nm=r['Name']; m=re.search(r'(k_[A-Za-z0-9_]+(?:<[^>]*>)?)',nm); nm=m.group(1) if m else nm[:60]
    print(f"{nm:50s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):6.2f} %")
PY
