"""Launch the K-C forward once per representative shape (target of rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsgcn_amd import native
lib = native.lib(); dev = 'cuda'; st = torch.cuda.current_stream().cuda_stream
n, V = 128, 25
for (Ci, Co, T) in [(64, 64, 64), (256, 256, 16)]:
    x1 = torch.randn(n, Ci, T, V, device=dev)
    s1 = torch.rand(Ci, device=dev) + .5; h1 = torch.randn(Ci, device=dev) * .1
    w = torch.randn(Co, Ci, device=dev) * Ci ** -.5; b = torch.zeros(Co, device=dev)
    z = torch.empty(n, Co, T, V, device=dev)
    part = torch.empty(4096, Co, 2, device=dev)
    for _ in range(3):
        assert lib.dsgcn_pwconv_fwd(x1.data_ptr(), s1.data_ptr(), h1.data_ptr(), None, None, None, 1, w.data_ptr(), b.data_ptr(),
                                    z.data_ptr(), None, part.data_ptr(), n, Ci, Co, T, V, 1, 0, 1, st) == 0
    torch.cuda.synchronize()
