#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r2q; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "pwconv or dynadj or units" > $O/test.log 2>&1
timeout 300 python tools/kc_bench.py > $O/kc.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc5 -o p --output-format csv -- python3 $R/tools/kc_once.py > $O/pmc5.log 2>&1
cd $R; python tools/pmc_summary.py $O/fetch.csv $O/pmc5 > /dev/null 2>&1; rm -rf $O/pmc5
tail -3 $O/test.log; cat $O/kc.log | cut -c1-66; cut -c1-200 $O/bench.json; cut -d, -f1,2,8,9 $O/fetch.csv
