#!/bin/bash
# L2 request volume per kernel of one eager step (TCC_HIT + TCC_MISS), next to the HBM-side figures of step_hbm_traffic.csv:
# which kernels ask L2 for far more bytes than they need from memory (re-read halos, per-tap reloads)?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_l2; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/raw -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-graph --no-roofline --no-cpu-baseline --no-other-configs > $O/run.log 2>&1
python3 - <<PY
import csv,glob,re,collections
tot=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for f in glob.glob('$O/raw/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f, newline='')):
        m=re.search(r'(k_[A-Za-z0-9_]+(?:<[^>]*>)?)', r['Kernel_Name']); k=m.group(1) if m else 'torch'
        tot[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
rows=sorted(tot.items(), key=lambda kv:-(kv[1].get('TCC_REQ_sum',0)))
with open('$O/l2_requests.csv','w') as fh:
    fh.write('kernel,launches,TCC_REQ_sum,TCC_HIT_sum,TCC_MISS_sum\n')
    for k,v in rows:
        fh.write(f"{k},{len(n[k])},{v.get('TCC_REQ_sum',0):.0f},{v.get('TCC_HIT_sum',0):.0f},{v.get('TCC_MISS_sum',0):.0f}\n")
for k,v in rows[:40]: print(f"{k:48s} {len(n[k]):4d} req {v.get('TCC_REQ_sum',0)/1e6:9.2f} M  hit {v.get('TCC_HIT_sum',0)/1e6:9.2f} M  miss {v.get('TCC_MISS_sum',0)/1e6:8.2f} M")
PY
rm -rf $O/raw
