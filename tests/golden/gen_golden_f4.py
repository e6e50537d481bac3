"""Golden fixtures for the f-4 backbones of SURVEY §8 — the original DG-STGCN spatial unit `dggcn` and the 2s-AGCN / AAGCN
unit `unit_aagcn` — generated from the IMPORTED reference (build container only: needs /root/reference):

    python tests/golden/gen_golden_f4.py

  unit_dggcn.npz              two units (64 -> 64 with a scalar alpha/beta, 64 -> 128 subset-wise), n = 2, T = 8, V = 25:
                              state_dict, input, output, input gradient and parameter gradients of an fp64 run
  model_reduced_dggcn.npz     DGSTGCN(gcn_type='dggcn', tcn_type='dgmstcn') at reduced width (base 16, 4 stages):
  model_reduced_dggcn_cfg.json logits / loss / gradients, fp64 truth and the reference's own fp32 run
  unit_aagcn.npz              two units (64 -> 64, 64 -> 128; adaptive + attention, all gates live), n = 2, T = 8, V = 25
  model_reduced_aagcn.npz     AAGCN (unit_aagcn + unit_tcn k=9, data_bn over M V C) at reduced width, as above

Data only (inputs, weights, the reference's outputs), no reference source."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (ref_shim, liven, reduced_model)

R = G.R


def unit_dggcn():
    np.random.seed(21)
    gr = R.graph.Graph(layout='nturgb+d', mode='random', num_filter=3, init_off=.04, init_std=.02)
    A = torch.tensor(gr.A, dtype=torch.float32)
    out = {}
    for i, (ci, co, sw) in enumerate([(64, 64, False), (64, 128, True)]):
        torch.manual_seed(300 + i)
        m = R.gutils.dggcn(ci, co, A, ratio=0.25, subset_wise=sw)
        G.liven(m, 27 + i)
        m64 = m.double()
        x = torch.randn(2, ci, 8, 25, dtype=torch.float64, requires_grad=True)
        Rm = torch.randn(2, co, 8, 25, dtype=torch.float64)
        y = m64(x)
        (y * Rm).sum().backward()
        tag = f'u{i}_'
        for k, v in m64.state_dict().items():
            out[tag + 'sd_' + k] = v.detach().numpy().astype(np.float32) if v.dtype.is_floating_point else v.numpy()
        out[tag + 'subset_wise'] = np.array(int(sw))
        out[tag + 'x'] = x.detach().numpy().astype(np.float32)
        out[tag + 'R'] = Rm.numpy().astype(np.float32)
        out[tag + 'y'] = y.detach().numpy().astype(np.float32)
        out[tag + 'dx'] = x.grad.numpy().astype(np.float32)
        for k, p in m64.named_parameters():
            if p.grad is not None and k in ('A', 'alpha', 'beta', 'conv1.weight', 'conv2.bias', 'pre.1.weight', 'post.weight'):
                out[tag + 'grad_' + k] = p.grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'unit_dggcn.npz'), **out)


def reduced_dggcn():
    cfg = dict(type='RecognizerGCN',
               backbone=dict(type='DGSTGCN', gcn_type='dggcn', gcn_ratio=0.25, gcn_ctr='T', gcn_ada='T', tcn_type='dgmstcn',
                             graph_cfg=dict(layout='nturgb+d', mode='random', num_filter=3, init_off=.04, init_std=.02),
                             tcn_ms_cfg=[(3, 1), (3, 2), (3, 3), (3, 4), ('max', 3), '1x1'],
                             base_channels=16, num_stages=4, inflate_stages=[3], down_stages=[3]),
               cls_head=dict(type='GCNHead', num_classes=12, in_channels=32))
    G.reduced_model(cfg, 'model_reduced_dggcn', seed=6)


AAGCN_GRADS = ('A', 'alpha', 'conv_d.1.weight', 'conv_a.0.weight', 'conv_b.2.bias', 'conv_ta.weight', 'conv_sa.weight',
               'fc1c.weight', 'fc2c.bias', 'bn.weight')


def liven_aagcn(m, seed):
    """The reference's init zeroes alpha, conv_ta and fc2c and sets bn.weight to 1e-6: make every path carry signal."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in m.modules():
            if type(mod).__name__ == 'unit_aagcn':
                mod.alpha.copy_(torch.randn(1, generator=g) * 0.5)
                mod.conv_ta.weight.copy_(torch.randn(mod.conv_ta.weight.shape, generator=g) * 0.1)
                mod.fc2c.weight.copy_(torch.randn(mod.fc2c.weight.shape, generator=g) * 0.05)
                mod.bn.weight.fill_(1.0)


def unit_aagcn():
    gr = R.graph.Graph(layout='nturgb+d', mode='spatial')
    A = torch.tensor(gr.A, dtype=torch.float32)
    out = {}
    for i, (ci, co) in enumerate([(64, 64), (64, 128)]):
        torch.manual_seed(400 + i)
        m = R.gutils.unit_aagcn(ci, co, A.clone())
        m.init_weights()
        liven_aagcn(m, 37 + i)
        m64 = m.double()
        x = torch.randn(2, ci, 8, 25, dtype=torch.float64, requires_grad=True)
        Rm = torch.randn(2, co, 8, 25, dtype=torch.float64)
        y = m64(x)
        (y * Rm).sum().backward()
        tag = f'u{i}_'
        for k, v in m64.state_dict().items():
            out[tag + 'sd_' + k] = v.detach().numpy().astype(np.float32) if v.dtype.is_floating_point else v.numpy()
        out[tag + 'x'] = x.detach().numpy().astype(np.float32)
        out[tag + 'R'] = Rm.numpy().astype(np.float32)
        out[tag + 'y'] = y.detach().numpy().astype(np.float32)
        out[tag + 'dx'] = x.grad.numpy().astype(np.float32)
        for k, p in m64.named_parameters():
            if p.grad is not None and k in AAGCN_GRADS:
                out[tag + 'grad_' + k] = p.grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'unit_aagcn.npz'), **out)


def reduced_aagcn():
    cfg = dict(type='RecognizerGCN',
               backbone=dict(type='AAGCN', graph_cfg=dict(layout='nturgb+d', mode='spatial'), base_channels=16, num_stages=4,
                             inflate_stages=[3], down_stages=[3]),
               cls_head=dict(type='GCNHead', num_classes=12, in_channels=32))
    liven0 = G.liven
    G.liven = lambda m, seed: liven_aagcn(m, seed)          # reduced_model() calls liven(): AAGCN has its own dead paths
    try:
        G.reduced_model(cfg, 'model_reduced_aagcn', seed=7)
    finally:
        G.liven = liven0


if __name__ == '__main__':
    unit_aagcn()
    reduced_aagcn()
    unit_dggcn()
    reduced_dggcn()
    print('wrote unit_dggcn.npz, model_reduced_dggcn.npz, unit_aagcn.npz, model_reduced_aagcn.npz')
