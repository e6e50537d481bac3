#!/bin/bash
# kernel-time profile of an arbitrary python command:  bash tools/gpu/prof_cmd.sh NAME TOPN DIVISOR -- python3 tools/x.py args
# writes gpurun_out/prof/NAME_kernel_stats.csv and prints the top kernels (times divided by DIVISOR = launches of the step)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof; mkdir -p $O
NAME=$1; TOP=$2; DIV=$3; shift; shift; shift; shift
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/raw_$NAME -o p --output-format csv -- "$@" > $O/$NAME.log 2>&1
f=$(find $O/raw_$NAME -name 'p_kernel_stats.csv' | head -1); cp "$f" $O/${NAME}_kernel_stats.csv; rm -rf $O/raw_$NAME
tail -1 $O/$NAME.log
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open('$O/${NAME}_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:$TOP]:
    nm=r['Name']; m=re.search(r'(k_[A-Za-z0-9_]+(?:<[^>]*>)?)',nm); nm=m.group(1) if m else nm[:60]
    print(f"{nm:52s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/$DIV/1e3:9.1f} us/step {float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):6.2f}")
print('total kernel ms/step', tot/$DIV/1e6)
PY
