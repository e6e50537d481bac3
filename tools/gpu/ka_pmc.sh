#!/bin/bash
# K-A counter passes (tools/ka_once.py under rocprofv3 --pmc), summary printed and kept in gpurun_out/ka_pmc/
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ka_pmc; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
P() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o p --output-format csv -- python3 $R/tools/ka_once.py > $O/$name.log 2>&1; }
P pmc1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
P pmc2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
P pmc3 TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
cd $R
python tools/pmc_summary.py $O/ka_pmc_summary.csv $O/pmc1 $O/pmc2 $O/pmc3 > $O/summary.log 2>&1
for d in pmc1 pmc2 pmc3; do rm -rf $O/$d; done
cat $O/ka_pmc_summary.csv
