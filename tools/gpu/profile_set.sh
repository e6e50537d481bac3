#!/bin/bash
# profile set of a round: bench line, rocprof stats of the bench command, K-C SQ counters, K-A and whole-step HBM traffic
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/profile_set_r6; mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
S() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats -d $O/raw_$name -o p --output-format csv -- "$@" > $O/$name.log 2>&1
      f=$(find $O/raw_$name -name 'p_kernel_stats.csv' | head -1); cp "$f" $O/$name.csv; rm -rf $O/raw_$name; }
S g_bench_cmd_kernel_stats python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-other-configs
S f_kernel_stats python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-graph --no-roofline --no-other-configs
P() { name=$1; prog=$2; shift; shift; timeout 400 rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o p --output-format csv -- python3 $R/$prog > $O/$name.log 2>&1; }
P pmc1 tools/kc_once.py SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
P pmc2 tools/kc_once.py SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD
P pmc3 tools/kc_once.py TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
P pmc4 tools/kc_once.py TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
P pmc5 tools/kc_once.py FETCH_SIZE
P pmc6 tools/kc_once.py WRITE_SIZE
P pmc7 tools/kc_once.py SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU
P kaf tools/ka_once.py FETCH_SIZE
P kaw tools/ka_once.py WRITE_SIZE
Q() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc $1 -d $O/$name -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-graph --no-roofline --no-cpu-baseline --no-other-configs > $O/$name.log 2>&1; }
Q stf FETCH_SIZE
Q stw WRITE_SIZE
cd $R
python tools/pmc_summary.py $O/kc_pmc_summary.csv $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5 $O/pmc6 $O/pmc7 > $O/summary.log 2>&1
python tools/ka_traffic.py $(find $O/kaf -name '*counter_collection.csv' | head -1) $(find $O/kaw -name '*counter_collection.csv' | head -1) $O/ka_traffic.json > $O/ka_traffic.log 2>&1
# 12 = 1 warm-up + 1 timed + the 10 host-enqueue steps bench.py runs
python tools/step_traffic.py $O/stf $O/stw 12 $O/step_hbm_traffic.csv > $O/step_traffic.log 2>&1
for d in pmc1 pmc2 pmc3 pmc4 pmc5 pmc6 pmc7 kaf kaw stf stw; do rm -rf $O/$d; done
cat $O/bench_n1.json | cut -c1-1500; cat $O/summary.log $O/ka_traffic.log $O/step_traffic.log; tail -2 $O/bench.err
# (round 6) the kernel sequences of one replayed step of the three BASELINE single-GPU configurations
for k in ds stgcn ctrgcn; do bash $R/tools/gpu/r6_seq.sh $k > $O/seq_$k.log 2>&1; cp $R/gpurun_out/r6_seq_$k/sequence.txt $O/step_sequence_$k.txt; done
cd /tmp
S stgcn_kernel_stats python3 $R/tools/bench_other.py stgcn 64 8
S ctrgcn_kernel_stats python3 $R/tools/bench_other.py ctrgcn 64 8
