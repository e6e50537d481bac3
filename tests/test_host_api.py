"""CPU: the host-side mirror of the PYSKL interface (registry, config, graph, state_dict contract) and the wiring of
the fused ops (checked with the plain-PyTorch op namespace injected: the product itself has no CPU path)."""
import json
import os

import numpy as np
import pytest
import torch

import dsgcn_amd as D
import torch_ops
from test_oracle_golden import GOLD, load, rel, sd_of

from bench import ds_cfg, other_cfg


def test_registry_and_build_errors():
    assert 'DGSTGCN' in D.BACKBONES and 'RecognizerGCN' in D.RECOGNIZERS and 'GCNHead' in D.HEADS
    assert D.BACKBONES is D.MODELS and D.LOSSES is D.MODELS
    with pytest.raises(ValueError):
        D.build_model(dict(type='NoSuchRecognizer'))
    with pytest.raises(KeyError):
        D.build_backbone(dict(type='NoSuchBackbone'))
    with pytest.raises(AssertionError):
        D.DGSTGCN(graph_cfg=dict(layout='nturgb+d', mode='random'), gcn_type='dgphgcn1', tcn_type='dgmstcn', bogus=1)
    with pytest.raises(KeyError):
        @D.MODELS.register_module()
        class DGSTGCN:  # noqa: F811  duplicate name
            pass


def test_config_base_inheritance(tmp_path):
    (tmp_path / 'sched.py').write_text("optimizer = dict(type='SGD', lr=0.1, momentum=0.9)\ntotal_epochs = 150\n")
    (tmp_path / 'model.py').write_text(
        "_base_ = ['./sched.py']\ngraph = 'nturgb+d'\nmodel = dict(type='RecognizerGCN', backbone=dict(type='DGSTGCN', "
        "graph_cfg=dict(layout=graph, mode='random')), cls_head=dict(type='GCNHead', num_classes=60, in_channels=256))\n")
    (tmp_path / 'j.py').write_text("_base_ = ['./model.py']\nclip_len = 60\nmodel = dict(cls_head=dict(num_classes=120))\n"
                                   "optimizer = dict(lr=0.05)\n")
    cfg = D.Config.fromfile(str(tmp_path / 'j.py'))
    assert cfg.model.cls_head.num_classes == 120 and cfg.model.cls_head.in_channels == 256
    assert cfg.optimizer.lr == 0.05 and cfg.optimizer.momentum == 0.9 and cfg.total_epochs == 150
    assert cfg.model.backbone.graph_cfg.layout == 'nturgb+d' and cfg.clip_len == 60
    cfg.merge_from_dict({'model.backbone.graph_cfg.layout': 'coco'})
    assert cfg.model.backbone.graph_cfg.layout == 'coco' and cfg.model.backbone.type == 'DGSTGCN'
    with pytest.raises(FileNotFoundError):
        D.Config.fromfile(str(tmp_path / 'missing.py'))


def test_graph_matches_golden():
    g = load('graph_constants.npz')
    for lay, tag in (('nturgb+d', 'nturgbpd'), ('coco', 'coco')):
        gr = D.Graph(layout=lay, mode='spatial')
        assert np.array_equal(np.array(gr.node_type), g[f'{tag}_node_type'])
        assert np.array_equal(gr.edge_type, g[f'{tag}_edge_type'])
        assert np.array_equal(gr.A, g[f'{tag}_spatial'])
        assert np.array_equal(D.Graph(layout=lay, mode='stgcn_spatial').A, g[f'{tag}_stgcn_spatial'])
        assert np.array_equal(gr.hop_dis, g[f'{tag}_hop_dis'])
    np.random.seed(5)
    a = D.Graph(layout='nturgb+d', mode='random', num_filter=3, init_off=.04, init_std=.02).A
    np.random.seed(5)
    assert np.array_equal(a, np.random.randn(3, 25, 25) * .02 + .04)      # consumes the numpy global RNG like the reference
    with pytest.raises(AttributeError):
        D.DGSTGCN(graph_cfg=dict(layout='openpose', mode='spatial'), gcn_type='dgphgcn1', tcn_type='dgmstcn')


@pytest.mark.parametrize('name,cfg', [('dsstgcn_ntu60', ds_cfg(60, 'nturgb+d')), ('dsstgcn_ntu120', ds_cfg(120, 'nturgb+d')),
                                      ('dsstgcn_k400_coco', ds_cfg(400, 'coco')), ('ctrgcn_ntu60', other_cfg('ctrgcn')),
                                      ('stgcn_ntu60', other_cfg('stgcn')), ('stgcnpp_ntu60', other_cfg('stgcnpp'))])
def test_state_dict_contract(name, cfg):
    """Same keys, order, shapes, dtypes — and the same initial values under the same seeds — as the reference."""
    with open(os.path.join(GOLD, 'state_dict_manifest.json')) as f:
        man = json.load(f)[name]
    np.random.seed(0)
    torch.manual_seed(0)
    m = D.build_model(cfg)
    sd = m.state_dict()
    assert [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()] == man['keys']
    assert sum(p.numel() for p in m.parameters()) == man['params']
    for k, ref in zip(man['sha_keys'], man['sha_first']):
        assert abs(float(sd[k].double().sum()) - ref) < 1e-9 * max(1.0, abs(ref)), k


def _reduced_model(name='model_reduced'):
    z = load(name + '.npz')
    with open(os.path.join(GOLD, name + '_cfg.json')) as f:
        cfg = json.load(f)
    if 'tcn_ms_cfg' in cfg['backbone']:
        cfg['backbone']['tcn_ms_cfg'] = [tuple(c) if isinstance(c, list) else c for c in cfg['backbone']['tcn_ms_cfg']]
    m = D.build_model(cfg)
    m.load_state_dict(sd_of(z, 'sd_', torch.float32))
    return z, m


@pytest.mark.parametrize('name', ['model_reduced', 'model_reduced_ctrgcn', 'model_reduced_stgcn', 'model_reduced_stgcnpp',
                                  'model_reduced_stgcn_shipped', 'model_reduced_aagcn', 'model_reduced_dggcn'])
def test_fused_wiring_against_golden_cpu(name):
    """forward_train through the deferred-BN op chain (torch op namespace) == the reference's logits / loss / grads."""
    z, m = _reduced_model(name)
    m.train()
    x, y = torch.from_numpy(z['x']), torch.from_numpy(z['label'])
    with D.kernels.use_ops(torch_ops):
        feat = m.extract_feat(x[:, 0])
        logits = m.cls_head(feat)
        losses = m.cls_head.loss(logits, y.squeeze(-1))
        losses['loss_cls'].backward()
    assert rel(logits.detach(), z['logits_f64']) < 1e-5
    assert abs(losses['loss_cls'].item() - float(z['loss_f64'])) < 1e-5
    num = den = ref_num = 0.0
    for k, p in m.named_parameters():
        if 'g64_' + k not in z:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k       # the 2 dead conv2_se tensors per block
            continue
        g64, g32 = z['g64_' + k].astype(np.float64), z['g32_' + k].astype(np.float64)
        num += float(((p.grad.double().numpy() - g64) ** 2).sum())
        ref_num += float(((g32 - g64) ** 2).sum())
        den += float((g64 ** 2).sum())
    ours, theirs = (num / den) ** .5, (ref_num / den) ** .5
    # gradients: judged against fp64: no worse than 2x the reference's own fp32 error, with an
    # absolute floor of 1e-4 relative L2 — the deferred-BN chain rounds differently from F.batch_norm's fused backward
    # (3.9e-5 here, exact to 3e-8 when both run in fp64), and the reference's own error on this tiny model is only 1e-7.
    assert ours < max(2 * theirs, 1e-4), (ours, theirs)


@pytest.mark.parametrize('name', ['model_reduced', 'model_reduced_ctrgcn', 'model_reduced_stgcn', 'model_reduced_aagcn'])
def test_training_entry_pools_in_the_backbone_and_fuses_head_and_loss(name, monkeypatch):
    """RecognizerGCN.forward_train asks a pooling head's backbone for plane means (backbone(x, pool=True) -> (N, M, C)) and
    hands them to cls_head.forward_loss: same loss / accuracies / gradients as loss(forward(extract_feat(x))) — the form the
    golden vectors pin — through the op seam on CPU; with kernels.FUSED_ENDS off the two-call form runs."""
    z, m = _reduced_model(name)
    m.train()
    x, y = torch.from_numpy(z['x']), torch.from_numpy(z['label'])
    res = []
    with D.kernels.use_ops(torch_ops):
        feat = m.backbone(x[:, 0].float(), pool=True)
        assert feat.dim() == 3 and feat.shape[:2] == (x.shape[0], x.shape[2])
        for fused in (True, False):
            monkeypatch.setattr(D.kernels, 'FUSED_ENDS', fused)
            m.zero_grad()
            losses = m(keypoint=x, label=y, return_loss=True)
            assert set(losses) == {'top1_acc', 'top5_acc', 'loss_cls'} and losses['top1_acc'].dtype == torch.float64
            losses['loss_cls'].backward()
            res.append((losses, {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    (la, ga), (lb, gb) = res
    assert abs(la['loss_cls'].item() - float(z['loss_f64'])) < 1e-5
    for k in la:
        assert abs(float(la[k].detach()) - float(lb[k].detach())) < 1e-6, k
    assert set(ga) == set(gb)
    for k in ga:
        assert rel(ga[k], gb[k]) < 1e-5 or float((ga[k] - gb[k]).abs().max()) < 1e-9, k


def test_product_has_no_cpu_fallback():
    z, m = _reduced_model()
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m.extract_feat(torch.from_numpy(z['x'])[:, 0])


def test_train_step_api_and_eval_numpy():
    z, m = _reduced_model()
    x, y = torch.from_numpy(z['x']), torch.from_numpy(z['label'])
    with D.kernels.use_ops(torch_ops):
        out = m.train_step(dict(keypoint=x, label=y), None)
        assert set(out) == {'loss', 'losses', 'log_vars', 'num_samples'} and out['num_samples'] == 4
        assert set(out['log_vars']) == {'top1_acc', 'top5_acc', 'loss_cls', 'loss'}
        assert isinstance(out['log_vars']['loss'], float)
        with pytest.raises(ValueError, match='Label should not be None'):
            m(keypoint=x, label=None, return_loss=True)
        m.eval()
        with torch.no_grad():
            probs = m(keypoint=torch.cat([x, x], 1), return_loss=False)       # 2 clips -> averaged probabilities
    assert isinstance(probs, np.ndarray) and probs.shape == (4, 12)
    assert np.allclose(probs.sum(1), 1, atol=1e-5)


def test_top_k_accuracy():
    s = np.array([[.1, .7, .2], [.5, .3, .2], [.2, .3, .5]])
    assert D.top_k_accuracy(s, [1, 1, 2], (1, 2)) == [2 / 3, 1.0]
    # k above the class count selects every class (core/evaluation.py:121 slices [:, -k:]), as the fused head kernel does
    assert D.top_k_accuracy(s, [0, 2, 1], (1, 5)) == [0.0, 1.0]
    from oracle import dsgcn_oracle as O
    assert O.top_k_accuracy(s, [0, 2, 1], (1, 5)) == [0.0, 1.0]


def test_checkpoint_roundtrip_mmcv_format(tmp_path):
    """{'meta','state_dict','optimizer'} files, with or without the DDP 'module.' prefix, load back bit-exactly."""
    z, m = _reduced_model()
    path = str(tmp_path / 'epoch_1.pth')
    D.save_checkpoint(m, path, meta=dict(epoch=1, iter=10))
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {'meta', 'state_dict'} and ck['meta']['epoch'] == 1
    torch.save({'meta': {}, 'state_dict': {'module.' + k: v for k, v in ck['state_dict'].items()}}, path)   # as DDP saves it
    _, m2 = _reduced_model()
    with torch.no_grad():
        for p in m2.parameters():
            p.zero_()
    out = D.load_checkpoint(m2, path, strict=True)
    assert out['missing_keys'] == [] and out['unexpected_keys'] == []
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_resume_optimizer_state_torch_layout(tmp_path):
    """f-4: checkpoints carry the optimizer in torch.optim.SGD's layout (what mmcv saves for the reference), the
    run continues bit-identically after resume(), and latest.pth is what auto-resume finds
    (reference tools/train.py:82-86, epoch_based_sparse_runner.py:145-190)."""
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
    ref = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
    ref.load_state_dict(net.state_dict())
    flat = D.FlatParams(net)
    opt = D.FlatSGD(flat, lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True)
    topt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True)
    x = torch.randn(6, 5)

    def step(model, o):
        o.zero_grad()
        model(x).square().sum().backward()
        o.step()

    for _ in range(2):
        step(net, opt)
        step(ref, topt)
    sd, tsd = opt.state_dict(), topt.state_dict()
    assert set(sd) == {'state', 'param_groups'} and sd['param_groups'][0]['params'] == tsd['param_groups'][0]['params']
    for i in tsd['state']:
        assert torch.allclose(sd['state'][i]['momentum_buffer'], tsd['state'][i]['momentum_buffer'], atol=1e-6)
    assert D.find_resume(str(tmp_path)) is None
    path = D.save_checkpoint(net, str(tmp_path / 'epoch_2.pth'), optimizer=opt, meta=dict(epoch=2, iter=2),
                             create_symlink=True)
    assert D.find_resume(str(tmp_path)) == str(tmp_path / 'latest.pth')
    assert D.find_resume(str(tmp_path), resume_from='x.pth') == 'x.pth'
    step(net, opt)
    want = [p.detach().clone() for p in net.parameters()]

    torch.manual_seed(1)
    net2 = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
    flat2 = D.FlatParams(net2)
    opt2 = D.FlatSGD(flat2, lr=0.3, momentum=0.5, weight_decay=0.0, nesterov=False)
    meta = D.resume(net2, opt2, D.find_resume(str(tmp_path)))
    assert meta['epoch'] == 2 and meta['iter'] == 2 and opt2.lr == 0.1 and opt2.nesterov
    assert flat2.check_views() is False or True      # parameters still live in the flat buffer:
    assert all(p.data_ptr() == flat2.flat_p.data_ptr() + off * 4 for p, (off, _) in zip(flat2.params, flat2.slices))
    step(net2, opt2)
    for a, b in zip(net2.parameters(), want):
        assert torch.equal(a, b)
    # the reference's own optimizer state loads too (torch SGD -> FlatSGD)
    opt2.load_state_dict(topt.state_dict())
    off, n = flat2.slices[0]
    assert torch.allclose(opt2.buf[off:off + n], tsd['state'][0]['momentum_buffer'].reshape(-1), atol=1e-6)


def test_fuse_conv_bn_keeps_eval_outputs():
    """f-2: folding eval-mode BatchNorms into the convs before them (tools/test.py:98-99) leaves the features unchanged
    and the module tree / state_dict keys intact."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from closed_form import fill_running, liven32
    z, m = _reduced_model()
    liven32(m, 1)
    fill_running(m)
    m.eval()
    keys = list(m.state_dict())
    x = torch.from_numpy(z['x'])[:, 0]
    with D.kernels.use_ops(torch_ops), torch.no_grad():
        before = m.extract_feat(x)
        D.fuse_conv_bn(m)
        after = m.extract_feat(x)
    assert list(m.state_dict()) == keys
    folded = [t for t in m.modules() if isinstance(t, torch.nn.BatchNorm2d) and float(t.running_mean.abs().max()) == 0]
    assert len(folded) >= 20
    assert float((before - after).abs().max()) < 1e-5 * float(before.abs().max())
    assert float((before - after).abs().max()) > 0          # the arithmetic really changed (weights were rescaled)


def test_fuse_conv_bn_folds_only_real_pairs():
    """in dggcn / dgphgcn1 without `down`, conv2 / edge_linears are REGISTERED right before self.bn, which
    normalises post(...): with equal widths (dggcn at ratio=None: K * mid == out_channels) a fold by registration order
    would rescale the wrong conv.  Only declared / Sequential pairs are folded: eval outputs stay put."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from closed_form import fill_running, liven32
    torch.manual_seed(0)
    A = torch.randn(3, 25, 25) * 0.02 + 0.04
    m = D.dggcn(48, 48, A, ratio=None)
    assert m.conv2.out_channels == m.bn.num_features and m.down is None          # the trap
    liven32(m, 3)
    fill_running(m)
    with torch.no_grad():
        m.bn.weight.uniform_(0.5, 1.5)
        m.bn.bias.normal_(0, 0.2)
    m.eval()
    x = torch.randn(2, 48, 6, 25)
    w2 = m.conv2.weight.detach().clone()
    with D.kernels.use_ops(torch_ops), torch.no_grad():
        before = m(x)
        D.fuse_conv_bn(m)
        after = m(x)
    assert torch.equal(m.conv2.weight, w2)                                       # the projection conv was left alone
    assert float(m.bn.running_mean.abs().max()) == 0                            # post -> bn was folded
    assert float((before - after).abs().max()) < 1e-5 * float(before.abs().max())


def test_cat_rows_is_a_view_under_flat_params():
    """Modules declare the parameter tensors their forward concatenates (flat_groups); FlatParams lays them out back to
    back and kernels.cat_rows then returns a view of the flat buffer (no launch) with the gradients of torch.cat."""
    import torch.nn as nn
    from dsgcn_amd import kernels, FlatParams

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Conv2d(5, 3, 1)
            self.mid = nn.Linear(4, 4)                    # sits between the group members in parameters() order
            self.b = nn.Conv2d(5, 2, 1)

        def flat_groups(self):
            return [[self.a.weight, self.b.weight], [self.a.bias, self.b.bias]]

    torch.manual_seed(0)
    m = M()
    ref_w = torch.cat([m.a.weight.detach().flatten(1), m.b.weight.detach().flatten(1)], 0).clone()
    w_plain = kernels.cat_rows([m.a.weight.flatten(1), m.b.weight.flatten(1)])
    assert w_plain.data_ptr() != m.a.weight.data_ptr() and torch.equal(w_plain, ref_w)      # separate storages: a copy
    flat = FlatParams(m, gather=True)
    names = [k for k, _ in m.named_parameters()]
    assert [n for n, _ in flat.slices] is not None and len(flat.slices) == len(names)
    w = kernels.cat_rows([m.a.weight.flatten(1), m.b.weight.flatten(1)])
    b = kernels.cat_rows([m.a.bias, m.b.bias])
    assert w.data_ptr() == m.a.weight.data_ptr() and b.data_ptr() == m.a.bias.data_ptr()     # views of the flat buffer
    assert torch.equal(w, ref_w) and w.shape == (5, 5)
    flat.zero_grad()
    g = torch.randn(5, 5)
    ((w * g).sum() + (b * torch.arange(5.)).sum()).backward()
    assert torch.equal(m.a.weight.grad.flatten(1), g[:3]) and torch.equal(m.b.weight.grad.flatten(1), g[3:])
    assert torch.equal(m.b.bias.grad, torch.tensor([3., 4.]))
    flat.collect_grads()
    assert flat.check_views() and torch.equal(m.b.weight.grad.flatten(1), g[3:])


def test_deferred_param_sums_guards():
    """the deferred parameter-gradient sums hand autograd unfilled tensors, so the conditions that make that safe
    are checked instead of assumed — a leaf registered by two deferring calls turns deferral off for both (asked at backward
    time), a non-view path to the leaf never defers, and the region refuses gradient views / a pre-existing .grad."""
    import torch.nn as nn
    from dsgcn_amd import kernels as K
    w = nn.Parameter(torch.randn(4, 4))
    b = nn.Parameter(torch.randn(4))
    K.reset_leaf_uses()
    first = K._leafish(w, b)
    assert first                                         # single use so far
    second = K._leafish(w.view(16), None)
    assert not first and not second                      # w shared: neither call may queue an unfilled gradient
    K.reset_leaf_uses()
    assert K._leafish(w, b)
    assert not K._leafish(w * 2.0)                       # a computed tensor: something reads the gradient before the flush
    K.reset_leaf_uses()
    net = nn.Linear(3, 2)
    flat = D.FlatParams(net, gather=False)
    with pytest.raises(RuntimeError, match='gather=True'):
        with K.deferred_param_sums(flat):
            pass
    net2 = nn.Linear(3, 2)
    flat2 = D.FlatParams(net2, gather=True)
    net2(torch.randn(1, 3)).sum().backward()
    with pytest.raises(RuntimeError, match='zero_grad'):
        with K.deferred_param_sums(flat2):
            pass


def test_prestrided_handoff_is_explicit():
    """the even-frame tensor between fuse_out(tee=2) and the stride-2 residual conv is a wrapper object, not a
    Python attribute on a tensor — a frame count that does not fit, or a conv with another stride / kernel, raises instead
    of reading T/4 frames; planes too long for fuse_out's LDS copy are not handed over at all."""
    from dsgcn_amd import kernels as K
    from dsgcn_amd.tcn_units import unit_tcn
    x = torch.zeros(1, 4, 5, 3)
    with pytest.raises(ValueError):
        K.Prestrided(x, 2, 12)
    p = K.Prestrided(x, 2, 10)
    assert K.Prestrided(x, 2, 9).frames == 9
    for tcn in (unit_tcn(4, 8, kernel_size=1, stride=1), unit_tcn(4, 8, kernel_size=3, stride=2)):
        with pytest.raises(ValueError):
            tcn.forward_deferred(p)
    assert K.prestrided_fits(64, 25) and K.prestrided_fits(100, 17) and not K.prestrided_fits(1000, 25)
