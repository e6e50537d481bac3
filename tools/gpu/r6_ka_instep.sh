#!/bin/bash
# K-A backward / forward launch geometry inside the replayed step: per variant the rocprof average of every K-A kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_ka_instep; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in "" "8=1024" "8=2048" "8=3072" "1=1536" "1=2560" "1=4096" "0=2048" "0=4096"; do
  n=$(echo "$v" | tr '=,' '__'); [ -z "$n" ] && n=default
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/raw_$n -o p --output-format csv -- python3 $R/tools/ka_instep.py "$v" > $O/$n.log 2>&1
  f=$(find $O/raw_$n -name 'p_kernel_stats.csv' | head -1)
  echo "== $n"
  python3 - "$f" <<'PY'
import csv, sys, re
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_aggregate' in r['Name']:
        name = re.search(r'(k_aggregate\w+<[^>]*>)', r['Name']).group(1)
        print(f"  {name:48s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs'])/1e3:7.2f} us  total {float(r['TotalDurationNs'])/1e3:9.1f} us")
        tot += float(r['TotalDurationNs']) / 1e3
print(f"  all K-A kernels: {tot:.1f} us")
PY
  rm -rf $O/raw_$n
done | tee $O/summary.txt
