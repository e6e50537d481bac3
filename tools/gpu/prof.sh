#!/bin/bash
# kernel-time profile of the bench step (eager), top kernels per step
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/raw -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-graph --no-roofline > $O/prof.log 2>&1
cp $O/raw/*/p_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || cp $O/raw/p_kernel_stats.csv $O/kernel_stats.csv
rm -rf $O/raw
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open('$O/kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:${1:-40}]:
    nm=r['Name']; m=re.search(r'(k_[A-Za-z0-9_]+(?:<[^>]*>)?)',nm); nm=m.group(1) if m else nm[:60]
    print(f"{nm:50s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/8/1e3:9.1f} us/step {float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):6.2f}")
print('total ms/step (8 steps)', tot/8/1e6)
PY
