"""Launch K-A forward and backward once per DS-STGCN layer shape (n=128) — target of the rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from dsgcn_amd import native
lib = native.lib(); dev = torch.device('cuda'); st = torch.cuda.current_stream().cuda_stream
V = 25
for rep in range(3):
    for (n, KC, T) in bench.ka_layer_shapes(128):
        zp = torch.randn(n, KC, T, V, device=dev); ah = torch.randn(n, KC, V, V, device=dev) * .2
        sc = torch.rand(KC, device=dev) + .5; sh = torch.randn(KC, device=dev) * .1
        y = torch.empty_like(zp); dy = torch.randn_like(zp); dzp = torch.empty_like(zp); dah = torch.empty_like(ah)
        part = torch.empty(4 * n * KC, 2, device=dev)
        assert lib.dsgcn_aggregate_fwd(zp.data_ptr(), sc.data_ptr(), sh.data_ptr(), 1, ah.data_ptr(), y.data_ptr(), n, KC, T, V, st) == 0
        assert lib.dsgcn_aggregate_bwd(zp.data_ptr(), sc.data_ptr(), sh.data_ptr(), 1, ah.data_ptr(), dy.data_ptr(), dzp.data_ptr(), dah.data_ptr(), part.data_ptr(), n, KC, T, V, st) == 0
    torch.cuda.synchronize()
print('alg bytes per 10-layer pass: fwd', sum(bench.ka_alg_bytes(n, KC, T, V, False) for n, KC, T in bench.ka_layer_shapes(128)),
      'bwd', sum(bench.ka_alg_bytes(n, KC, T, V, True) for n, KC, T in bench.ka_layer_shapes(128)))
