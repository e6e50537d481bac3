"""Generate the golden fixtures from the IMPORTED reference (build container only: needs /root/reference).

    python tests/golden/gen_golden.py

Writes small .npz/.json files next to this script.  Each fixture holds inputs (weights, activations) and the
reference's outputs for them — data only, no reference source.  fp64 runs give the "truth"; the fp32 reference
outputs are stored as well where the noise floor matters (gradients).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

R = ref_shim.load()


def ds_cfg(num_classes=60, layout='nturgb+d', **bk):
    backbone = dict(
        type='DGSTGCN', gcn_type='dgphgcn1', gcn_ratio=0.125, gcn_node_attention=True, gcn_edge_attention=True,
        gcn_decompose=True, gcn_subset_wise=True, gcn_ctr='T', gcn_ada='T', tcn_type='dgmstcn',
        graph_cfg=dict(layout=layout, mode='random', num_filter=3, init_off=.04, init_std=.02),
        tcn_ms_cfg=[(3, 1), (3, 2), (3, 3), (3, 4), ('max', 3), '1x1'])
    backbone.update(bk)
    return dict(type='RecognizerGCN', backbone=backbone,
                cls_head=dict(type='GCNHead', num_classes=num_classes, in_channels=bk.get('_head_in', 256)))


def liven(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, p in module.named_parameters():
            if k.endswith(('alpha', 'beta', 'add_coeff')):
                p.copy_(torch.randn(p.shape, generator=g, dtype=torch.float64).to(p.dtype) * 0.5)


def sd_np(module, prefix=''):
    return {prefix + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def graphs():
    out = {}
    for lay in ('nturgb+d', 'coco'):
        g = R.graph.Graph(layout=lay, mode='spatial')
        tag = lay.replace('+', 'p')
        out[f'{tag}_node_type'] = np.array(g.node_type)
        out[f'{tag}_edge_type'] = g.edge_type
        out[f'{tag}_spatial'] = g.A
        out[f'{tag}_stgcn_spatial'] = R.graph.Graph(layout=lay, mode='stgcn_spatial').A
        out[f'{tag}_hop_dis'] = g.hop_dis
    np.savez_compressed(os.path.join(HERE, 'graph_constants.npz'), **out)


def unit_dgphgcn1():
    np.random.seed(11)
    G = R.graph.Graph(layout='nturgb+d', mode='random', num_filter=3, init_off=.04, init_std=.02)
    A = torch.tensor(G.A, dtype=torch.float32)
    nt = torch.tensor(G.node_type)
    et = torch.tensor(G.edge_type, dtype=torch.float32)
    out = {}
    for i, (ci, co) in enumerate([(3, 64), (64, 64), (64, 128)]):
        torch.manual_seed(100 + i)
        m = R.gutils.dgphgcn1(ci, co, A, et, nt, ratio=0.125, decompose=True, node_attention=True,
                              edge_attention=True, subset_wise=True, ctr='T', ada='T')
        liven(m, 7 + i)
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
            if p.grad is not None and k in ('A', 'alpha', 'beta', 'edge_linears.weight', 'conv1_se.weight', 'pre.1.weight'):
                out[tag + 'grad_' + k] = p.grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'unit_dgphgcn1.npz'), **out)


def unit_dgmstcn():
    out = {}
    for i, (c, s) in enumerate([(64, 1), (96, 2)]):
        torch.manual_seed(200 + i)
        m = R.gutils.dgmstcn(c, c, stride=s)
        liven(m, 17 + i)
        m64 = m.double()
        x = torch.randn(2, c, 8, 25, dtype=torch.float64, requires_grad=True)
        y = m64(x)
        Rm = torch.randn(y.shape, dtype=torch.float64)
        (y * Rm).sum().backward()
        tag = f't{i}_'
        for k, v in m64.state_dict().items():
            out[tag + 'sd_' + k] = v.detach().numpy().astype(np.float32) if v.dtype.is_floating_point else v.numpy()
        out[tag + 'x'] = x.detach().numpy().astype(np.float32)
        out[tag + 'R'] = Rm.numpy().astype(np.float32)
        out[tag + 'y'] = y.detach().numpy().astype(np.float32)
        out[tag + 'dx'] = x.grad.numpy().astype(np.float32)
        out[tag + 'grad_add_coeff'] = m64.add_coeff.grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'unit_dgmstcn.npz'), **out)


def other_cfg(kind, num_classes=60, **bk):
    if kind == 'ctrgcn':
        backbone = dict(type='CTRGCN', gcn_type='unit_ctrgcn', graph_cfg=dict(layout='nturgb+d', mode='spatial'))
    elif kind == 'stgcn_shipped':      # configs/stgcn/STGCN_model.py
        backbone = dict(type='STGCN', gcn_adaptive='init', tcn_type='unitmlp', tcn_add_tcn=True, tcn_merge_after=True,
                        graph_cfg=dict(layout='nturgb+d', mode='random', num_filter=3, init_off=.04, init_std=.02))
    elif kind == 'stgcnpp':
        backbone = dict(type='STGCN', gcn_adaptive='init', gcn_with_res=True, tcn_type='mstcn',
                        graph_cfg=dict(layout='nturgb+d', mode='spatial'))
    else:
        backbone = dict(type='STGCN', graph_cfg=dict(layout='nturgb+d', mode='stgcn_spatial'))
    backbone.update(bk)
    return dict(type='RecognizerGCN', backbone=backbone,
                cls_head=dict(type='GCNHead', num_classes=num_classes, in_channels=256))


def extract_feat_f64(m64, x):
    """The reference's STGCN.forward casts its input to fp32 (stgcn.py:141); for the fp64 truth run its own data_bn and
    blocks on the fp64 input directly (same modules, same order)."""
    bb = m64.backbone
    if type(bb).__name__ != 'STGCN':
        return m64.extract_feat(x)
    N, M, T, V, C = x.size()
    h = x.permute(0, 1, 3, 4, 2).contiguous()
    h = bb.data_bn(h.view(N * M, V * C, T))
    h = h.view(N, M, V, C, T).permute(0, 1, 3, 4, 2).contiguous().view(N * M, C, T, V)
    for i in range(bb.num_stages):
        h = bb.gcn[i](h)
    return h.reshape((N, M) + h.shape[1:])


def reduced_model(cfg=None, name='model_reduced', seed=3):
    if cfg is None:
        cfg = ds_cfg(num_classes=12, base_channels=16, num_stages=4, inflate_stages=[3], down_stages=[3])
        cfg['cls_head']['in_channels'] = 32
    np.random.seed(seed)
    torch.manual_seed(seed)
    m = R.builder.build_model(cfg)
    liven(m, 33)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 1, 2, 16, 25, 3, generator=g)
    y = torch.randint(0, 12, (4, 1), generator=g)
    out = {'sd_' + k: v for k, v in sd_np(m).items()}
    out['x'] = x.numpy()
    out['label'] = y.numpy()
    # fp32 reference run (what the reference itself produces)
    feat = m.extract_feat(x[:, 0])
    logits = m.cls_head(feat)
    loss = m.cls_head.loss(logits, y.squeeze(-1))['loss_cls']
    loss.backward()
    out['logits_f32'] = logits.detach().numpy()
    out['loss_f32'] = np.array(loss.item())
    for k, p in m.named_parameters():
        if p.grad is not None:
            out['g32_' + k] = p.grad.numpy()
    # fp64 truth
    m64 = R.builder.build_model(cfg).double()
    m64.load_state_dict({k: v.double() if v.dtype.is_floating_point else v for k, v in m.state_dict().items()})
    feat = extract_feat_f64(m64, x[:, 0].double())
    logits64 = m64.cls_head(feat)
    loss64 = torch.nn.functional.cross_entropy(logits64, y.squeeze(-1))
    loss64.backward()
    out['logits_f64'] = logits64.detach().numpy().astype(np.float32)
    out['loss_f64'] = np.array(loss64.item())
    for k, p in m64.named_parameters():
        if p.grad is not None:
            out['g64_' + k] = p.grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    with open(os.path.join(HERE, name + '_cfg.json'), 'w') as f:
        json.dump(cfg, f, indent=1)


def reduced_other_models():
    for kind in ('ctrgcn', 'stgcn', 'stgcnpp'):
        cfg = other_cfg(kind, 12, base_channels=16, num_stages=4, inflate_stages=[3], down_stages=[3])
        cfg['cls_head']['in_channels'] = 32
        reduced_model(cfg, 'model_reduced_' + kind, seed=4)


from closed_form import closed_form_fill, counter_input  # noqa: E402


def full_size():
    """G4: full-width models with closed-form weights and inputs — only logits, loss and per-tensor gradient norms are
    stored (the weights / inputs are regenerated by the same formulas in tests/test_model_gpu.py)."""
    out = {}
    cases = [('dsstgcn_ntu60', ds_cfg(60), 64, 25), ('dsstgcn_k400_coco', ds_cfg(400, 'coco'), 100, 17),
             ('ctrgcn_ntu60', other_cfg('ctrgcn'), 64, 25), ('stgcnpp_ntu60', other_cfg('stgcnpp'), 64, 25)]
    for name, cfg, T, V in cases:
        np.random.seed(0)
        torch.manual_seed(0)
        m = R.builder.build_model(cfg)
        closed_form_fill(m)
        classes = cfg['cls_head']['num_classes']
        x, y = counter_input(2, T, V, classes)
        m.train()
        logits = m.cls_head(m.extract_feat(x[:, 0]))
        loss = torch.nn.functional.cross_entropy(logits, y.squeeze(-1))
        loss.backward()
        out[name + '_logits'] = logits.detach().numpy()
        out[name + '_loss'] = np.array(loss.item())
        names = [k for k, p in m.named_parameters() if p.grad is not None]
        out[name + '_gnorm'] = np.array([float(p.grad.double().norm()) for k, p in m.named_parameters() if p.grad is not None])
        with open(os.path.join(HERE, f'full_{name}_gradnames.json'), 'w') as f:
            json.dump(names, f)
        # fp64 truth of the same case (the fp32 gradients of a 10-block train-mode-BN network are only ~1e-3..1e-1
        # accurate per tensor: the tests judge against fp64, relative to the reference's own fp32 error)
        m64 = R.builder.build_model(cfg).double()
        m64.load_state_dict({k: v.double() if v.dtype.is_floating_point else v for k, v in m.state_dict().items()})
        m64.train()
        logits64 = m64.cls_head(extract_feat_f64(m64, x[:, 0].double()))
        loss64 = torch.nn.functional.cross_entropy(logits64, y.squeeze(-1))
        loss64.backward()
        g64 = dict(m64.named_parameters())
        out[name + '_logits64'] = logits64.detach().numpy().astype(np.float32)
        out[name + '_loss64'] = np.array(loss64.item())
        out[name + '_gnorm64'] = np.array([float(g64[k].grad.norm()) for k in names])
        # per-tensor relative error of the reference's fp32 gradient against fp64 (full tensors compared here)
        g32 = dict(m.named_parameters())
        out[name + '_gerr32'] = np.array([float((g32[k].grad.double() - g64[k].grad).norm() / (g64[k].grad.norm() + 1e-30))
                                          for k in names])
        tot = (sum(float((g32[k].grad.double() - g64[k].grad).pow(2).sum()) for k in names) /
               sum(float(g64[k].grad.pow(2).sum()) for k in names)) ** .5
        out[name + '_gerr32_total'] = np.array(tot)
        print(name, 'ref fp32 grad error vs fp64 (whole gradient):', tot)
        print(name, float(loss), logits.abs().mean().item())
    np.savez_compressed(os.path.join(HERE, 'full_size.npz'), **out)


def manifests():
    man = {}
    ds_keys = ('backbone.gcn.0.gcn.A', 'backbone.gcn.0.gcn.pre.0.weight', 'backbone.gcn.9.tcn.transform.2.weight',
               'cls_head.fc_cls.weight')
    ctr_keys = ('backbone.net.0.gcn1.convs.0.conv1.weight', 'backbone.net.4.gcn1.convs.2.conv4.weight',
                'backbone.net.9.tcn1.branches.1.3.conv.weight', 'cls_head.fc_cls.weight')
    st_keys = ('backbone.gcn.0.gcn.conv.weight', 'backbone.gcn.4.residual.conv.weight',
               'backbone.gcn.9.tcn.conv.weight', 'cls_head.fc_cls.weight')
    for name, cfg, keys in (('dsstgcn_ntu60', ds_cfg(60), ds_keys), ('dsstgcn_ntu120', ds_cfg(120), ds_keys),
                            ('dsstgcn_k400_coco', ds_cfg(400, 'coco'), ds_keys),
                            ('ctrgcn_ntu60', other_cfg('ctrgcn'), ctr_keys), ('stgcn_ntu60', other_cfg('stgcn'), st_keys),
                            ('stgcnpp_ntu60', other_cfg('stgcnpp'), ('backbone.gcn.0.gcn.conv.weight',
                                                                      'backbone.gcn.4.gcn.down.0.weight',
                                                                      'backbone.gcn.9.tcn.transform.2.weight',
                                                                      'cls_head.fc_cls.weight'))):
        np.random.seed(0)
        torch.manual_seed(0)
        m = R.builder.build_model(cfg)
        man[name] = dict(keys=[[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()],
                         params=sum(p.numel() for p in m.parameters()), sha_keys=list(keys),
                         sha_first=[float(m.state_dict()[k].double().sum()) for k in keys])
    with open(os.path.join(HERE, 'state_dict_manifest.json'), 'w') as f:
        json.dump(man, f)


if __name__ == '__main__':
    graphs()
    unit_dgphgcn1()
    unit_dgmstcn()
    reduced_model()
    reduced_other_models()
    full_size()
    manifests()
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(('.npz', '.json')):
            print(fn, os.path.getsize(os.path.join(HERE, fn)))
