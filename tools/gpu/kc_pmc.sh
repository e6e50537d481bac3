#!/bin/bash
# K-C counter passes only (tools/kc_once.py under rocprofv3 --pmc), summary printed and kept in gpurun_out/kc_pmc/
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/kc_pmc; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
P() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o p --output-format csv -- python3 $R/tools/kc_once.py > $O/$name.log 2>&1; }
P pmc1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
P pmc2 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
P pmc3 TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
P pmc4 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
P pmc5 FETCH_SIZE
cd $R
python tools/pmc_summary.py $O/kc_pmc_summary.csv $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5 > $O/summary.log 2>&1
for d in pmc1 pmc2 pmc3 pmc4 pmc5; do rm -rf $O/$d; done
cat $O/kc_pmc_summary.csv
