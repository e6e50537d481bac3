#!/bin/bash
# round 2, call A: baseline numbers + SQ counters of the K-C kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2a; mkdir -p $O
cd $R
timeout 300 python tools/pw_bench.py > $O/pw_bench.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 120 rocprofv3 -L > $O/counters.txt 2>&1
P() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o p --output-format csv -- python3 $R/tools/kc_once.py > $O/$name.log 2>&1; }
P pmc1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
P pmc2 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD
P pmc3 TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
P pmc4 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
P pmc5 FETCH_SIZE
P pmc6 WRITE_SIZE
P pmc7 SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU
cd $R
python tools/pmc_summary.py $O/kc_pmc_summary.csv $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5 $O/pmc6 $O/pmc7 > $O/summary.log 2>&1
# keep only the summary + logs (raw csvs are large)
for d in pmc1 pmc2 pmc3 pmc4 pmc5 pmc6 pmc7; do rm -rf $O/$d; done
tail -3 $O/pw_bench.log; cat $O/bench.json | cut -c1-400; cat $O/summary.log
