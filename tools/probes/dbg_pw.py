import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import torch
from dsgcn_amd import kernels as K
import torch_ops as R
for (n, Ci, Co, T, V) in [(1, 3, 24, 7, 25), (2, 16, 24, 7, 25), (2, 8, 64, 25, 25), (3, 32, 48, 3, 25), (2, 16, 24, 9, 25), (2,16,24,6,25)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, Ci, T, V, generator=g); w = torch.randn(Co, Ci, 1, 1, generator=g); b = torch.randn(Co, generator=g)
    z = K.pwconv(x.cuda(), None, None, None, False, w.cuda(), b.cuda(), 1, False)[0].cpu()
    zr = R.pwconv(x.double(), None, None, None, False, w.double(), b.double(), 1, False)[0]
    d = (z.double() - zr).abs()
    bad = (d > 1e-4).nonzero()
    L = T * V
    print((n, Ci, Co, T, V), 'L', L, 'L%4', L % 4, 'bad', len(bad), 'of', z.numel())
    if len(bad):
        flat = sorted(set((int(i[0]), int(i[2]) * V + int(i[3])) for i in bad))
        print('   bad (n, pos):', flat[:40])
        print('   bad channels:', sorted(set(int(i[1]) for i in bad))[:30])
