"""Launch the K-C forward / data-gradient / weight-gradient a few times per representative shape
(target program of the rocprofv3 --pmc passes whose summaries live under profiles/)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsgcn_amd import native
lib = native.lib(); dev = 'cuda'; st = torch.cuda.current_stream().cuda_stream
n, V = 128, 25
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for (Ci, Co, T) in [(64, 64, 64), (256, 256, 16)]:
    x1 = torch.randn(n, Ci, T, V, device=dev)
    s1 = torch.rand(Ci, device=dev) + .5; h1 = torch.randn(Ci, device=dev) * .1
    w = torch.randn(Co, Ci, device=dev) * Ci ** -.5; b = torch.zeros(Co, device=dev)
    z = torch.empty(n, Co, T, V, device=dev)
    gz = torch.randn(n, Co, T, V, device=dev)
    A0 = torch.randn(Co, device=dev) * 1e-3; B0 = torch.randn(Co, device=dev) * 1e-3
    dx = torch.empty_like(x1)
    part = torch.empty(lib.dsgcn_pwconv_partial_rows(n, Ci, Co, T, V, 1, 0), Co, 2, device=dev)
    ipart = torch.empty(lib.dsgcn_pwconv_ipart_rows(n, Ci, Co, T, V, 1), Ci, 3, device=dev)
    splits = lib.dsgcn_pwconv_wgrad_splits(n, Ci, Co, T, V, 1)
    pstride = Co * Ci + Co
    wpart = torch.empty(splits, pstride, device=dev)
    wsb = lib.dsgcn_pwconv_wsplit_bytes(n, Ci, Co, T, V, 1)      # the wide convs run with their pre-split weight image
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    if wsb:
        assert lib.dsgcn_pwconv_wsplit(w.data_ptr(), Ci, Co, ws.data_ptr(), st) == 0
    wsp = ws.data_ptr() if wsb else None
    for _ in range(reps):
        assert lib.dsgcn_pwconv_fwd_ws(x1.data_ptr(), s1.data_ptr(), h1.data_ptr(), None, None, None, 1, w.data_ptr(),
                                    b.data_ptr(), z.data_ptr(), None, part.data_ptr(), n, Ci, Co, T, V, 1, 0, 1, wsp, st) == 0
        assert lib.dsgcn_pwconv_dgrad_ws(x1.data_ptr(), s1.data_ptr(), h1.data_ptr(), None, None, None, 1, w.data_ptr(),
                                      z.data_ptr(), None, gz.data_ptr(), None, A0.data_ptr(), B0.data_ptr(),
                                      dx.data_ptr(), None, ipart.data_ptr(), n, Ci, Co, T, V, 1, 0, wsp, st) == 0
        assert lib.dsgcn_pwconv_wgrad(x1.data_ptr(), s1.data_ptr(), h1.data_ptr(), None, None, None, 1, z.data_ptr(),
                                      None, gz.data_ptr(), None, A0.data_ptr(), B0.data_ptr(), wpart.data_ptr(),
                                      wpart.data_ptr() + 4 * Co * Ci, pstride, n, Ci, Co, T, V, 1, 0, st) == 0
    torch.cuda.synchronize()
