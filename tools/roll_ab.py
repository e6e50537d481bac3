"""A/B of tuning switches on the whole step: python tools/roll_ab.py <mask> [nograph]
mask bits 0-3: dsgcn_pwconv_tuning key 2 (forward roll, dgrad roll, dgrad epilogue prefetch, XCD-aware wgrad).
(Weight gradients on a second stream were tried as bit 4: 30 % slower eagerly (record_stream bookkeeping, allocator
churn) and the capture of that fork pattern aborted; removed.)"""
import os, sys, runpy
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
mask = int(sys.argv[1])
extra = ['--no-graph'] if len(sys.argv) > 2 and sys.argv[2] == 'nograph' else []
sys.argv = ['bench.py', '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-roofline'] + extra
from dsgcn_amd import native
native.lib().dsgcn_pwconv_tuning(2, mask & 15)
runpy.run_path(os.path.join(R, 'bench.py'), run_name='__main__')
