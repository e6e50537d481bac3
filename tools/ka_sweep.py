"""K-A launch-geometry sweep over the model's 10-layer shape mix (lab library knobs; same timing as bench.py's roofline)."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dsgcn_amd import native
lib = native.lib() if os.environ.get('KA_LIB') == 'product' else native.lab_lib(); dev = torch.device('cuda'); st = torch.cuda.current_stream().cuda_stream
V = 25; n = 128
bufs = []
for (nn, KC, t) in bench.ka_layer_shapes(n):
    zp = torch.randn(nn, KC, t, V, device=dev); ah = torch.randn(nn, KC, V, V, device=dev) * .2
    bufs.append(dict(zp=zp, ah=ah, sc=torch.rand(KC, device=dev) + .5, sh=torch.randn(KC, device=dev) * .1, y=torch.empty_like(zp),
                     dy=torch.randn_like(zp), dzp=torch.empty_like(zp), dah=torch.empty_like(ah),
                     part=torch.empty(4 * nn * KC, 2, device=dev), dims=(nn, KC, t)))
def fwd(b):
    nn, KC, t = b['dims']
    assert lib.dsgcn_aggregate_fwd(b['zp'].data_ptr(), b['sc'].data_ptr(), b['sh'].data_ptr(), 1, b['ah'].data_ptr(), b['y'].data_ptr(), nn, KC, t, V, st) == 0
def bwd(b):
    nn, KC, t = b['dims']
    assert lib.dsgcn_aggregate_bwd(b['zp'].data_ptr(), b['sc'].data_ptr(), b['sh'].data_ptr(), 1, b['ah'].data_ptr(), b['dy'].data_ptr(), b['dzp'].data_ptr(), b['dah'].data_ptr(), b['part'].data_ptr(), nn, KC, t, V, st) == 0
def timeit(fn, reps=10):
    for b in bufs: fn(b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for b in bufs: fn(b)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
fb = sum(bench.ka_alg_bytes(*b['dims'], V, False) for b in bufs); bb = sum(bench.ka_alg_bytes(*b['dims'], V, True) for b in bufs)
for _ in range(int(os.environ.get('KA_REPEAT', 1))):
    print(os.environ.get('KA_LIB', 'lab'), 'default fwd %.0f GB/s  bwd %.0f GB/s' % (fb / timeit(fwd, 20) / 1e3, bb / timeit(bwd, 20) / 1e3), flush=True)
if os.environ.get('KA_SWEEP', '1') == '0':
    sys.exit(0)
for waves, chunk, direct in itertools.product((1024, 2048, 3072, 4096, 6144, 8192), (32, 64), (0, 1)):
    lib.dsgcn_set_tuning(0, waves); lib.dsgcn_set_tuning(5, chunk); lib.dsgcn_set_tuning(4, direct)
    print(f'fwd waves {waves} chunk {chunk} direct {direct}: {fb / timeit(fwd) / 1e3:.0f} GB/s', flush=True)
lib.dsgcn_set_tuning(0, 0); lib.dsgcn_set_tuning(5, 32); lib.dsgcn_set_tuning(4, 1)
for pw, wgs, pair in itertools.product((1024, 2048, 3072, 4096), (768, 1024, 1536, 2048, 3072), (0, 1)):
    if pair == 0 and wgs != 768: continue
    lib.dsgcn_set_tuning(1, pw); lib.dsgcn_set_tuning(8, wgs); lib.dsgcn_set_tuning(7, pair)
    print(f'bwd pipe-waves {pw} pair {pair} pair-wgs {wgs}: {bb / timeit(bwd) / 1e3:.0f} GB/s', flush=True)
