#!/bin/bash
# L2 request volume per kernel for another configuration:  bash tools/gpu/r5_l2_other.sh ctrgcn|stgcn
R=${GRAFT_REPO_ROOT:-/root/repo}; K=${1:-ctrgcn}; O=$R/gpurun_out/r5_l2_$K; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
DSGCN_EAGER=1 timeout 600 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $O/raw -o p --output-format csv -- python3 $R/tools/bench_other.py $K 64 1 > $O/run.log 2>&1
python3 - <<PY
import csv,glob,re,collections
tot=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set); dur=collections.defaultdict(float)
for f in glob.glob('$O/raw/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f, newline='')):
        m=re.search(r'(k_[A-Za-z0-9_]+(?:<[^>]*>)?)', r['Kernel_Name']); k=m.group(1) if m else 'torch'
        tot[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
rows=sorted(tot.items(), key=lambda kv:-(kv[1].get('TCC_REQ_sum',0)))
steps=7.0   # 3 warm-up + 3 + 1 timed eager steps of bench_other
with open('$O/l2_requests.csv','w') as fh:
    fh.write('kernel,launches_per_step,TCC_REQ_M_per_step,hit_share\n')
    for k,v in rows:
        fh.write(f"{k},{len(n[k])/steps:.1f},{v.get('TCC_REQ_sum',0)/1e6/steps:.2f},{v.get('TCC_HIT_sum',0)/max(v.get('TCC_REQ_sum',1),1):.2f}\n")
for k,v in rows[:28]: print(f"{k:48s} {len(n[k])/steps:5.1f} req/step {v.get('TCC_REQ_sum',0)/1e6/steps:9.2f} M  (~{v.get('TCC_REQ_sum',0)/steps*110/1e9:6.2f} GB)  hit {v.get('TCC_HIT_sum',0)/max(v.get('TCC_REQ_sum',1),1):.2f}")
PY
rm -rf $O/raw
