#!/bin/bash
# SQ counters of every kernel of the step (two passes over tools/step_once.py, eager): matrix-pipe busy cycles, waits, instruction mix
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/step_pmc; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
P() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o p --output-format csv -- python3 $R/tools/step_once.py 3 > $O/$name.log 2>&1; }
P pmc1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
P pmc2 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
P pmc3 GRBM_GUI_ACTIVE
cd $R
python tools/pmc_summary.py $O/step_pmc_summary.csv $O/pmc1 $O/pmc2 $O/pmc3 > $O/summary.log 2>&1
for d in pmc1 pmc2 pmc3; do rm -rf $O/$d; done
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/step_pmc_summary.csv')))
print(f"{'kernel':44s} {'grid':>8s} {'n':>3s} {'us':>7s} {'mfma_busy%':>10s} {'wait%':>6s} {'valu/mfma':>9s}")
for r in sorted(rows, key=lambda r: -float(r['avg_us'])*int(r['dispatches']))[:45]:
    us=float(r['avg_us']); cyc=us*1e-6*float(r.get('GRBM_GUI_ACTIVE',0) or 0)/max(us*1e-6,1e-12)
    gui=float(r.get('GRBM_GUI_ACTIVE',0) or 0)/8.0          # cycles of the launch (sum over 8 XCDs)
    busy=float(r.get('SQ_VALU_MFMA_BUSY_CYCLES',0) or 0)
    wc=float(r.get('SQ_WAVE_CYCLES',0) or 1); wa=float(r.get('SQ_WAIT_ANY',0) or 0)
    mf=float(r.get('SQ_INSTS_MFMA',0) or 0); va=float(r.get('SQ_INSTS_VALU',0) or 0)
    pct=busy/(gui*1024) *100 if gui else 0
    print(f"{r['kernel'][:44]:44s} {r['grid']:>8s} {r['dispatches']:>3s} {us:7.1f} {pct:10.1f} {wa/wc*100:6.1f} {va/max(mf,1):9.1f}")
PY
