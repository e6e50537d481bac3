import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from dsgcn_amd import native
lib = native.lib(); dev = torch.device('cuda'); st = torch.cuda.current_stream().cuda_stream
V, n = 25, 128
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for KC, T in [(24, 64), (96, 16)]:
    sets = []
    for _ in range(6):
        zp = torch.randn(n, KC, T, V, device=dev); ah = torch.randn(n, KC, V, V, device=dev) * .2
        sets.append((zp, ah, torch.empty_like(zp), torch.randn_like(zp), torch.empty_like(zp), torch.empty_like(ah), torch.empty(4 * n * KC, 2, device=dev)))
    sc = torch.rand(KC, device=dev) + .5; sh = torch.randn(KC, device=dev) * .1
    idx = [0]
    def bwd():
        zp, ah, y, dy, dzp, dah, part = sets[idx[0] % 6]; idx[0] += 1
        assert lib.dsgcn_aggregate_bwd(zp.data_ptr(), sc.data_ptr(), sh.data_ptr(), 1, ah.data_ptr(), dy.data_ptr(), dzp.data_ptr(), dah.data_ptr(), part.data_ptr(), n, KC, T, V, st) == 0
    for w in (2048, 3072):
        lib.dsgcn_set_tuning(1, w)
        line = f'KC={KC} T={T} waves={w}: '
        for ab, nm in ((0, 'full'), (1, '-dA_mfma'), (2, '-dP'), (3, '-allmfma'), (4, '-dAstore'), (8, '-dzpstore'), (15, 'loads_only')):
            lib.dsgcn_set_tuning(3, ab)
            line += f'{nm} {timeit(bwd):6.1f} | '
        lib.dsgcn_set_tuning(3, 0)
        print(line, flush=True)
