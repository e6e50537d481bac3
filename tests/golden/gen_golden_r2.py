"""Round-2 golden fixtures from the IMPORTED reference (build container only: needs /root/reference).

    python tests/golden/gen_golden_r2.py

Adds to gen_golden.py's set (which it leaves untouched):
  full_grads_<config>.npz   full-width models, DEFAULT init (seed 0) + live alpha/beta/add_coeff, closed-form input (8 clips):
                            reference logits / loss (fp32 and fp64) and the full fp64 gradients of a fixed selection of
                            ~12 tensors + the reference's own fp32 error on that selection (a well-conditioned case,
                            unlike full_size.npz's sine weights: a tight full-width gradient check);
  eval_<config>.npz         eval-mode (running-statistics BatchNorm) test-time scores: RecognizerGCN.forward_test on
                            2 samples x 10 clips, softmax-averaged (recognizergcn.py:53-107);
  unit_intermediates.npz    dgphgcn1 units (V=25 and V=17) with the INTERMEDIATES of the reference's forward — xbar,
                            x1, x2, tanh(D), softmax(G), Ahat, P, Y — captured from the reference's own torch.einsum /
                            conv calls, fp64;
  unit_others.npz           unit_gcn, unit_tcn (k=9; k=1 stride 2), unit_ctrgcn, MSTCN at real widths: state_dict, input,
                            output, input gradient and parameter gradients (fp64 run, stored fp32).
Data only (inputs and reference outputs); no reference source."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from closed_form import (UNIT_CASES, calibrate_running, counter_clips, counter_input, eval_clips, EVAL_LIVEN, fill_running, liven32, make_unit, pick_tensors,  # noqa: E402
                         sd_digest)
from gen_golden import ds_cfg, extract_feat_f64, other_cfg  # noqa: E402

R = ref_shim.load()

def shipped_ctr_cfg(num_classes=60, **bk):
    """configs/ctrgcn/CTRGCN_model.py: unit_ctrhgcn + msmlp on the random graph."""
    backbone = dict(type='CTRGCN', gcn_type='unit_ctrhgcn', gcn_node_attention=True, gcn_edge_attention=True,
                    gcn_add_type=False, gcn_ada=True, gcn_num_types=5, gcn_rel_reduction=8, gcn_edge_num=15,
                    tcn_type='msmlp', tcn_add_tcn=True, tcn_merge_after=True,
                    graph_cfg=dict(layout='nturgb+d', mode='random', num_filter=3, init_off=.04, init_std=.02))
    backbone.update(bk)
    return dict(type='RecognizerGCN', backbone=backbone,
                cls_head=dict(type='GCNHead', num_classes=num_classes, in_channels=256))


CONFIGS = {
    'dsstgcn_ntu60': (lambda: ds_cfg(60), 64, 25),            # BASELINE config 2
    'dsstgcn_ntu120': (lambda: ds_cfg(120), 64, 25),          # BASELINE config 3
    'dsstgcn_k400_coco': (lambda: ds_cfg(400, 'coco'), 100, 17),   # BASELINE config 5
    'ctrgcn_ntu60': (lambda: other_cfg('ctrgcn'), 64, 25),    # BASELINE config 4
    'stgcn_ntu60': (lambda: other_cfg('stgcn'), 64, 25),      # BASELINE config 1
    'stgcnpp_ntu60': (lambda: other_cfg('stgcnpp'), 64, 25),
    'ctrgcn_shipped_ntu60': (lambda: shipped_ctr_cfg(), 64, 25),       # the CTR-GCN variant the reference ships (f-1)
    'stgcn_shipped_ntu60': (lambda: other_cfg('stgcn_shipped'), 64, 25),    # the ST-GCN variant the reference ships (unitmlp)
}
GRAD_CLIPS = 8           # clips per full_grads case (16 person-samples of batch statistics)


def build(cfg, scale=0.5):
    np.random.seed(0)
    torch.manual_seed(0)
    m = R.builder.build_model(cfg)
    liven32(m, 1, scale)
    return m


def to64(m, cfg):
    m64 = R.builder.build_model(cfg).double()
    m64.load_state_dict({k: v.double() if v.dtype.is_floating_point else v for k, v in m.state_dict().items()})
    return m64


def full_grads():
    for name, (mk, T, V) in CONFIGS.items():
        cfg = mk()
        classes = cfg['cls_head']['num_classes']
        m = build(cfg).train()
        if name == 'stgcn_ntu60':
            for mod in m.modules():                     # vanilla ST-GCN's Dropout(0.5) would make the case random
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
        x, y = counter_input(GRAD_CLIPS, T, V, classes)
        logits = m.cls_head(m.extract_feat(x[:, 0]))
        loss = torch.nn.functional.cross_entropy(logits, y.squeeze(-1))
        loss.backward()
        m64 = to64(m, cfg).train()
        for mod in m64.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        logits64 = m64.cls_head(extract_feat_f64(m64, x[:, 0].double()))
        loss64 = torch.nn.functional.cross_entropy(logits64, y.squeeze(-1))
        loss64.backward()
        g32 = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
        g64 = {k: p.grad for k, p in m64.named_parameters() if p.grad is not None}
        names = pick_tensors([(k, g.numel()) for k, g in g64.items()])
        out = dict(names=np.array(json.dumps(names)), logits32=logits.detach().numpy(), loss32=np.array(loss.item()),
                   logits64=logits64.detach().numpy().astype(np.float32), loss64=np.array(loss64.item()))
        num = den = 0.0
        for i, k in enumerate(names):
            out[f'g64_{i}'] = g64[k].numpy().astype(np.float32)
            num += float((g32[k].double() - g64[k]).pow(2).sum())
            den += float(g64[k].pow(2).sum())
        out['gerr32_set'] = np.array((num / den) ** .5)
        allnum = sum(float((g32[k].double() - g64[k]).pow(2).sum()) for k in g64)
        allden = sum(float(g64[k].pow(2).sum()) for k in g64)
        out['gerr32_total'] = np.array((allnum / allden) ** .5)
        # conditioning of the case: the same fp32 reference run on an input perturbed at the 1e-7 level (below fp32
        # resolution of most entries' neighbours: what any change of summation order amounts to).  Train-mode BN over 4
        # person-samples + ReLU / max-pool decisions amplify it; the distance between the two fp32 gradients is the
        # noise floor no fp32 implementation can be expected to beat.
        m2 = build(cfg).train()
        for mod in m2.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        sign = torch.where(torch.arange(x.numel()).reshape(x.shape) % 2 == 0, 1.0, -1.0)
        x2 = (x.double() * (1.0 + 2e-7 * sign)).float()
        torch.nn.functional.cross_entropy(m2.cls_head(m2.extract_feat(x2[:, 0])), y.squeeze(-1)).backward()
        g2 = {k: p.grad for k, p in m2.named_parameters() if p.grad is not None}
        nnum = sum(float((g2[k].double() - g32[k].double()).pow(2).sum()) for k in names)
        out['gnoise32_set'] = np.array((nnum / den) ** .5)
        # BatchNorm running statistics after this one training forward (fp32 run): first / middle / last layers
        rk = [k for k in m.state_dict() if k.endswith(('running_mean', 'running_var'))]
        rk = [rk[i] for i in sorted({0, 1, len(rk) // 2, len(rk) // 2 + 1, len(rk) - 2, len(rk) - 1})]
        out['running_names'] = np.array(json.dumps(rk))
        for i, k in enumerate(rk):
            out[f'running_{i}'] = m.state_dict()[k].numpy()
        np.savez_compressed(os.path.join(HERE, f'full_grads_{name}.npz'), **out)
        print(name, 'noise', float(out['gnoise32_set']), 'ref fp32 gradient error vs fp64: selection', float(out['gerr32_set']), 'whole', float(out['gerr32_total']),
              'logits', float((logits.double() - logits64).norm() / logits64.norm()))


def eval_fixtures():
    for name, (mk, T, V) in CONFIGS.items():
        cfg = mk()
        m = build(cfg, EVAL_LIVEN.get(name, 0.5))
        x = eval_clips(name, T, V)
        calibrate_running(m, x[:, 0], lambda mod, inp: mod.extract_feat(inp))
        with torch.no_grad():
            probs = m(keypoint=x, return_loss=False)
        m64 = to64(m, cfg).eval()
        with torch.no_grad():
            feat = torch.cat([extract_feat_f64(m64, x[:, c].double()) for c in range(x.shape[1])])   # clip-major
            sc = m64.cls_head(feat).reshape(x.shape[1], 2, -1).permute(1, 0, 2)
            clip_probs = torch.softmax(sc, 2)
            probs64 = clip_probs.mean(1).numpy()
        out = dict(probs32=np.asarray(probs, dtype=np.float32), probs64=probs64.astype(np.float32),
                   probs64_clips=clip_probs.numpy().astype(np.float32), scores64_clips=sc.numpy().astype(np.float32))
        rk = [k for k in m.state_dict() if k.endswith(('running_mean', 'running_var'))]
        out['running_keys'] = np.array(json.dumps(rk))
        out['running_values'] = np.concatenate([m.state_dict()[k].numpy().reshape(-1) for k in rk]).astype(np.float32)
        np.savez_compressed(os.path.join(HERE, f'eval_{name}.npz'), **out)
        assert np.isfinite(probs64).all(), name
        print(name, 'eval: ref fp32 vs fp64', float(np.abs(probs - probs64).max()), 'max prob', float(probs64.max()))


class EinsumTap:
    """Records every torch.einsum call (equation, operands, result) made while active."""

    def __enter__(self):
        self.calls = []
        self._orig = torch.einsum

        def tapped(eq, *ops):
            res = self._orig(eq, *ops)
            self.calls.append((eq, [o.detach().clone() for o in ops], res.detach().clone()))
            return res
        torch.einsum = tapped
        return self

    def __exit__(self, *exc):
        torch.einsum = self._orig
        return False


def unit_intermediates():
    out = {}
    for tag, layout, V, ci, co, seed in (('v25', 'nturgb+d', 25, 64, 64, 21), ('v17', 'coco', 17, 64, 128, 22)):
        np.random.seed(seed)
        G = R.graph.Graph(layout=layout, mode='random', num_filter=3, init_off=.04, init_std=.02)
        A = torch.tensor(G.A, dtype=torch.float32)
        torch.manual_seed(seed)
        m = R.gutils.dgphgcn1(ci, co, A, torch.tensor(G.edge_type, dtype=torch.float32), torch.tensor(G.node_type),
                              ratio=0.125, decompose=True, node_attention=True, edge_attention=True, subset_wise=True,
                              ctr='T', ada='T')
        liven32(m, seed)
        m = m.double().train()
        x = torch.randn(2, ci, 8, V, dtype=torch.float64)
        grabbed = {}
        h = m.conv1.register_forward_hook(lambda mod, inp, outp: grabbed.__setitem__('xbar', inp[0].detach().clone()))
        with EinsumTap() as tap:
            y = m(x)
        h.remove()
        by_eq = {}
        for eq, ops, res in tap.calls:
            by_eq.setdefault(eq, []).append((ops, res))
        (ops_g, _), = by_eq['nkctv,nkctw->nktvw']
        (ops_y, res_y), = by_eq['nkctv,nkcvw->nkctw']
        scal = by_eq['nkctuv,k->nkctuv']                      # [tanh(D) * alpha, softmax(G) * beta]
        assert len(scal) == 2
        f32 = lambda t: t.numpy().astype(np.float32)          # noqa: E731
        for k, v in m.state_dict().items():
            out[f'{tag}_sd_{k}'] = f32(v) if v.dtype.is_floating_point else v.numpy()
        out[f'{tag}_node_type'] = np.array(G.node_type)
        out[f'{tag}_edge_type'] = np.array(G.edge_type)
        out[f'{tag}_x'] = f32(x)
        out[f'{tag}_xbar'] = f32(grabbed['xbar'][:, :, 0])            # (n, Ci, V)
        out[f'{tag}_x1'] = f32(ops_g[0][:, :, :, 0])                  # (n, K, mid, V)
        out[f'{tag}_x2'] = f32(ops_g[1][:, :, :, 0])
        out[f'{tag}_tanhD'] = f32(scal[0][0][0][:, :, :, 0])          # (n, K, mid, V, V)
        out[f'{tag}_softG'] = f32(scal[1][0][0][:, :, 0, 0])          # (n, K, V, V)
        out[f'{tag}_Ahat'] = f32(ops_y[1])                            # (n, K, mid, V, V)
        out[f'{tag}_P'] = f32(ops_y[0])                               # (n, K, mid, T, V)  after pre's BN + ReLU
        out[f'{tag}_Y'] = f32(res_y)
        out[f'{tag}_out'] = f32(y.detach())
    np.savez_compressed(os.path.join(HERE, 'unit_intermediates.npz'), **out)


def unit_others():
    """Units rebuilt on both sides from the same seeded recipe (closed_form.make_unit: weights are not stored, their
    digest is): reference output, input gradient and parameter gradients from an fp64 run."""
    out = {}
    Gs = R.graph.Graph(layout='nturgb+d', mode='spatial')
    A = torch.tensor(Gs.A, dtype=torch.float32)
    for tag in UNIT_CASES:
        m, x, Rm = make_unit(R.gutils, tag, A, Gs.edge_type, Gs.node_type)
        out[f'{tag}_digest'] = np.array(sd_digest(m))
        m = m.double().train()
        x = x.double().requires_grad_()
        y = m(x)
        (y * Rm.double()).sum().backward()
        f32 = lambda t: t.detach().numpy().astype(np.float32)  # noqa: E731
        out[f'{tag}_y'] = f32(y)
        out[f'{tag}_dx'] = f32(x.grad)
        for k, p in m.named_parameters():
            if p.grad is not None:
                if p.numel() <= 16384:
                    out[f'{tag}_grad_{k}'] = f32(p.grad)
                else:
                    out[f'{tag}_gnorm_{k}'] = np.array(float(p.grad.norm()))
        for k, v in m.state_dict().items():
            if 'running' in k:
                out[f'{tag}_{k}'] = f32(v)
    np.savez_compressed(os.path.join(HERE, 'unit_others.npz'), **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['units', 'others', 'grads', 'eval']
    if 'units' in which:
        unit_intermediates()
    if 'others' in which:
        unit_others()
    if 'grads' in which:
        full_grads()
    if 'eval' in which:
        eval_fixtures()
    for fn in sorted(os.listdir(HERE)):
        if fn.startswith(('full_grads_', 'eval_', 'unit_intermediates', 'unit_others')):
            print(fn, os.path.getsize(os.path.join(HERE, fn)))
