import os, sys, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests')); sys.path.insert(0, os.path.join(R, 'tests', 'golden'))
import numpy as np, torch
import dsgcn_amd as D, torch_ops
from bench import ds_cfg
from closed_form import closed_form_fill, counter_input
z = np.load(os.path.join(R, 'tests/golden/full_size.npz'))
def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64); return np.linalg.norm(a - b) / np.linalg.norm(b)
for name, cfg, T, V, cls in [('dsstgcn_k400_coco', ds_cfg(400, 'coco'), 100, 17, 400), ('dsstgcn_ntu60', ds_cfg(60, 'nturgb+d'), 64, 25, 60)]:
    np.random.seed(0); torch.manual_seed(0)
    m = D.build_model(cfg); closed_form_fill(m)
    x, y = counter_input(2, T, V, cls)
    m = m.cuda().train()
    with torch.no_grad():
        lk = m.cls_head(m.extract_feat(x.cuda()[:, 0])).cpu()
        with D.kernels.use_ops(torch_ops):
            lt = m.cls_head(m.extract_feat(x.cuda()[:, 0])).cpu()
            m64 = m.double()
            l64 = m64.cls_head(m64.extract_feat(x.cuda().double()[:, 0])).cpu()
    print(name, 'hip vs ref64', rel(lk, z[name + '_logits64']), '| torch_ops fp32 (same folded-BN wiring) vs ref64', rel(lt, z[name + '_logits64']),
          '| torch_ops fp64 vs ref64', rel(l64, z[name + '_logits64']), '| ref32 vs ref64', rel(z[name + '_logits'], z[name + '_logits64']))
