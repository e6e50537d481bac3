"""Per-layer timing of the K-A (gather-aggregate) variants on the GPU box (A/B evidence for DESIGN.md)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dsgcn_amd import native

lib = native.lab_lib()
dev = torch.device('cuda')
st = torch.cuda.current_stream().cuda_stream
V = 25
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
shapes = [(24, 64), (48, 64), (48, 32), (96, 32), (96, 16)]


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


for KC, T in shapes:
    # several buffer sets so consecutive launches do not hit the same lines in the Infinity Cache
    sets = []
    for _ in range(6):
        zp = torch.randn(n, KC, T, V, device=dev); ah = torch.randn(n, KC, V, V, device=dev) * .2
        sets.append((zp, ah, torch.empty_like(zp), torch.randn_like(zp), torch.empty_like(zp), torch.empty_like(ah),
                     torch.empty(4 * n * KC, 2, device=dev)))
    sc = torch.rand(KC, device=dev) + .5; sh = torch.randn(KC, device=dev) * .1
    fb = bench.ka_alg_bytes(n, KC, T, V, False); bb = bench.ka_alg_bytes(n, KC, T, V, True)
    idx = [0]

    def mk(variant):
        def f():
            zp, ah, y = sets[idx[0] % 6][:3]; idx[0] += 1
            rc = lib.dsgcn_aggregate_fwd_variant(zp.data_ptr(), sc.data_ptr(), sh.data_ptr(), 1, ah.data_ptr(), y.data_ptr(),
                                                 n, KC, T, V, variant, st)
            assert rc == 0
        return f

    def bwd():
        zp, ah, y, dy, dzp, dah, part = sets[idx[0] % 6]; idx[0] += 1
        rc = lib.dsgcn_aggregate_bwd(zp.data_ptr(), sc.data_ptr(), sh.data_ptr(), 1, ah.data_ptr(), dy.data_ptr(),
                                     dzp.data_ptr(), dah.data_ptr(), part.data_ptr(), n, KC, T, V, st)
        assert rc == 0

    def cp():
        zp, ah, y = sets[idx[0] % 6][:3]; idx[0] += 1
        y.copy_(zp)

    line = f'KC={KC:3d} T={T:3d} fwdMB={fb/1e6:6.1f} '
    t = timeit(cp); line += f'| copy {t:6.1f}us {2*zp.numel()*4/t/1e3:6.0f}GB/s '
    for v, nm in ():
        t = timeit(mk(v)); line += f'| {nm} {t:6.1f}us {fb/t/1e3:6.0f}GB/s '
    for w in (3072, 4096, 6144):
        lib.dsgcn_set_tuning(0, w); lib.dsgcn_set_tuning(5, 32)
        t = timeit(mk(0)); line += f'| fwd w{w} {t:5.1f}us {fb/t/1e3:5.0f} '
    lib.dsgcn_set_tuning(0, 0)
    for hr, ws in ((0, (2048, 3072, 4096)), (1, (1536, 2048, 3072))):
        lib.dsgcn_set_tuning(7, hr)
        for w in ws:
            lib.dsgcn_set_tuning(8 if hr else 1, w)
            t = timeit(bwd); line += f'| bwd {"pair" if hr else "pipe"} w{w} {t:5.1f}us {bb/t/1e3:5.0f}'
    lib.dsgcn_set_tuning(8, 0); lib.dsgcn_set_tuning(1, 0); lib.dsgcn_set_tuning(7, 1)
    print(line, flush=True)
