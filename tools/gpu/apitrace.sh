R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/api; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --hip-runtime-trace --memory-copy-trace --stats -d $O/raw -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-graph --no-roofline > $O/log.txt 2>&1
ls $O/raw/* | head; for f in $(find $O/raw -name '*stats*.csv'); do echo == $f; head -25 $f; done
rm -rf $O/raw
