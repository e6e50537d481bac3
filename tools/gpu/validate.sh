#!/bin/bash
# full GPU validation: every -m gpu test, smoke(), default bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/validate; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -q -m gpu > $O/test.log 2>&1
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -4 $O/test.log; tail -3 $O/smoke.log; cut -c1-400 $O/bench.json
