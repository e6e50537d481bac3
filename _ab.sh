#!/bin/bash
cd $GRAFT_REPO_ROOT
B() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
echo nt; B; B; B
sed -i 's/constexpr int P4_NT_STORE = 2; /constexpr int P4_NT_STORE = 0; /' ds-gcn_amd/csrc/pw4.hip
python -c "
import sys; sys.path.insert(0,'.')
import dsgcn_amd.native as n; n.build()" > /dev/null 2>&1
echo plain; B; B; B
