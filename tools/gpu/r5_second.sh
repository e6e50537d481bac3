#!/bin/bash
# round 5: the full-size parity cases + the new bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_second; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "aggregate or dynadj or pwconv or tconv_gemm or temporal_ms" > $O/test_k.log 2>&1; tail -30 $O/test_k.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r5_second/bench.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], {k:v['frac'] for k,v in d['roofline_other'].items()})
print(d['roofline_step']); print(d['other_configs'])
PY
tail -3 $O/bench.err
