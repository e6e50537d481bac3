"""A/B of the K-C rolling prefetch on the whole step: python tools/roll_ab.py <mask> (bit 0 forward, bit 1 dgrad)"""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mask = int(sys.argv[1])
sys.argv = ['bench.py', '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-roofline']
from dsgcn_amd import native
native.lib().dsgcn_pwconv_tuning(2, mask)
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'), run_name='__main__')
