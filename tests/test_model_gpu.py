"""-m gpu: the product path (HIP kernels through the C ABI) at model level against the committed golden vectors and the
CPU oracle.  Bar (BASELINE.json north_star): logits and loss within 1e-4 relative, fp32."""
import json
import os

import numpy as np
import pytest
import torch

import dsgcn_amd as D
from dsgcn_amd import native
from oracle import dsgcn_oracle as O
from bench import build_model, ds_cfg, other_cfg
from test_oracle_golden import GOLD, load, rel, sd_of

pytestmark = pytest.mark.gpu


def test_native_library_is_loaded():
    lib = native.lib()
    assert lib.dsgcn_version() >= 100
    with open('/proc/self/maps') as f:
        assert 'libdsgcn.so' in f.read()           # the in-tree HIP library really is the code that runs


@pytest.mark.parametrize('name', ['model_reduced', 'model_reduced_ctrgcn', 'model_reduced_stgcn', 'model_reduced_stgcnpp',
                                  'model_reduced_dggcn', 'model_reduced_aagcn', 'model_reduced_stgcn_shipped'])
def test_reduced_model_vs_golden(name):
    """DS-STGCN, classic CTR-GCN, ST-GCN / ST-GCN++ the original DG-STGCN (gcn_type='dggcn') and AAGCN at reduced widths against
    the reference's committed outputs."""
    z = load(name + '.npz')
    with open(os.path.join(GOLD, name + '_cfg.json')) as f:
        cfg = json.load(f)
    if 'tcn_ms_cfg' in cfg['backbone']:
        cfg['backbone']['tcn_ms_cfg'] = [tuple(c) if isinstance(c, list) else c for c in cfg['backbone']['tcn_ms_cfg']]
    m = D.build_model(cfg)
    m.load_state_dict(sd_of(z, 'sd_', torch.float32))
    m = m.cuda().train()
    x, y = torch.from_numpy(z['x']).cuda(), torch.from_numpy(z['label']).cuda()
    losses = m(keypoint=x, label=y, return_loss=True)
    feat = m.extract_feat(x[:, 0])
    logits = m.cls_head(feat)
    loss = m.cls_head.loss(logits, y.squeeze(-1))['loss_cls']
    loss.backward()
    assert rel(logits.detach().cpu(), z['logits_f64']) < 1e-4
    assert abs(loss.item() - float(z['loss_f64'])) / abs(float(z['loss_f64'])) < 1e-4
    assert abs(losses['loss_cls'].item() - loss.item()) < 1e-5
    num = den = 0.0
    for k, p in m.named_parameters():
        if 'g64_' + k in z:
            g64 = z['g64_' + k].astype(np.float64)
            num += float(((p.grad.double().cpu().numpy() - g64) ** 2).sum())
            den += float((g64 ** 2).sum())
    assert (num / den) ** .5 < 2e-4, (num / den) ** .5        # whole-gradient relative L2 vs fp64 truth


@pytest.mark.parametrize('layout,V,T,classes', [('nturgb+d', 25, 64, 60), ('nturgb+d', 25, 64, 120), ('coco', 17, 100, 400)])
def test_full_model_vs_oracle(layout, V, T, classes):
    """Full-width DS-STGCN — BASELINE config 2 (NTU-60), config 3 (NTU-120, 120 classes) and config 5 (K400, V=17,
    T=100, 400 classes) — 2 clips, against the CPU oracle."""
    np.random.seed(0)
    torch.manual_seed(0)
    m = D.build_model(ds_cfg(classes, layout))
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith(('alpha', 'beta', 'add_coeff')):
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(2, 1, 2, T, V, 3, generator=g)
    y = torch.randint(0, classes, (2, 1), generator=g)
    gc = O.graph_constants(layout)
    ref_logits, ref_loss = O.recognizer_forward_train(x, y, sd, gc['node_type'], gc['edge_type'], O.dgstgcn_plan())
    m = m.cuda().train()
    out = m.train_step(dict(keypoint=x.cuda(), label=y.cuda()), None)
    feat = m.extract_feat(x.cuda()[:, 0])
    logits = m.cls_head(feat)
    assert rel(logits.detach().cpu(), ref_logits) < 1e-4
    assert abs(out['log_vars']['loss'] - ref_loss.item()) / abs(ref_loss.item()) < 1e-4
    out['loss'].backward()
    dead = [k for k, p in m.named_parameters() if p.grad is None]
    assert len(dead) == 20 and all('conv2_se' in k for k in dead)          # reference quirk Q1


@pytest.mark.parametrize('kind', ['ctrgcn', 'stgcn', 'stgcnpp', 'ctrgcn_shipped', 'stgcn_shipped'])
def test_full_other_backbones_vs_oracle(kind):
    """Full-width classic CTR-GCN (BASELINE config 4) and vanilla ST-GCN (config 1), 2 clips, against the CPU oracle."""
    np.random.seed(0)
    torch.manual_seed(0)
    m = D.build_model(other_cfg(kind))
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith('alpha'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(2, 1, 2, 64, 25, 3, generator=g)
    y = torch.randint(0, 60, (2, 1), generator=g)
    plan = O.ctrgcn_plan() if kind.startswith('ctrgcn') else O.dgstgcn_plan()
    ref_logits, ref_loss = O.recognizer_forward_train_backbone(kind, x, y, sd, plan)
    m = m.cuda().train()
    out = m.train_step(dict(keypoint=x.cuda(), label=y.cuda()), None)
    logits = m.cls_head(m.extract_feat(x.cuda()[:, 0]))
    assert rel(logits.detach().cpu(), ref_logits) < 1e-4
    assert abs(out['log_vars']['loss'] - ref_loss.item()) / abs(ref_loss.item()) < 1e-4
    out['loss'].backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


@pytest.mark.parametrize('name,T,V,classes', [('dsstgcn_ntu60', 64, 25, 60), ('dsstgcn_k400_coco', 100, 17, 400),
                                              ('ctrgcn_ntu60', 64, 25, 60), ('stgcnpp_ntu60', 64, 25, 60)])
def test_full_size_vs_reference_fixture(name, T, V, classes):
    """Full-width models (BASELINE configs 2, 5, 4 and ST-GCN++) with closed-form weights and inputs against the outputs
    the REFERENCE produced for them (tests/golden/full_size.npz, its fp32 and fp64 runs)."""
    import sys
    sys.path.insert(0, GOLD)
    from closed_form import closed_form_fill, counter_input
    cfg = {'dsstgcn_ntu60': ds_cfg(60, 'nturgb+d'), 'dsstgcn_k400_coco': ds_cfg(400, 'coco'),
           'ctrgcn_ntu60': other_cfg('ctrgcn'), 'stgcnpp_ntu60': other_cfg('stgcnpp')}[name]
    np.random.seed(0)
    torch.manual_seed(0)
    m = D.build_model(cfg)
    closed_form_fill(m)
    z = load('full_size.npz')
    with open(os.path.join(GOLD, f'full_{name}_gradnames.json')) as f:
        names = json.load(f)
    x, y = counter_input(2, T, V, classes)
    m = m.cuda().train()
    logits = m.cls_head(m.extract_feat(x.cuda()[:, 0]))
    loss = torch.nn.functional.cross_entropy(logits, y.cuda().squeeze(-1))
    loss.backward()
    # judged against the reference's fp64 evaluation of the same case; the reference's own fp32 run is the yardstick:
    # logits / loss within max(2x its error, 1e-4) (north_star bar 1e-4)
    l64 = z[name + '_logits64']
    ref_err = rel(z[name + '_logits'], l64)
    assert rel(logits.detach().cpu(), l64) < max(2 * ref_err, 1e-4), (rel(logits.detach().cpu(), l64), ref_err)
    assert abs(loss.item() - float(z[name + '_loss64'])) / abs(float(z[name + '_loss64'])) < 1e-4
    # (gradients at full width are checked tensor by tensor in test_full_width_gradients_vs_reference_fixture; with these
    # sine weights the reference's own fp32 gradient is 3.5-25 % off its fp64 one, so this case pins logits / loss only)
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in m.parameters())


@pytest.mark.parametrize('kind,T,N', [('ds', 18, 3), ('ds', 7, 1), ('ds', 33, 2), ('ctrgcn', 18, 3), ('stgcnpp', 9, 2),
                                      ('stgcn', 21, 1)])
def test_ragged_shapes_vs_oracle(kind, T, N):
    """Frame counts that are not multiples of 4 / of the stride (the non-vectorised and remainder paths of every kernel),
    single-clip batches: reduced-width models against the CPU oracle, logits and loss 1e-4."""
    if kind == 'ds':
        cfg = ds_cfg(12, 'nturgb+d')
        cfg['backbone'].update(base_channels=16, num_stages=4, inflate_stages=[3], down_stages=[3])
    else:
        cfg = other_cfg(kind, 12, base_channels=16, num_stages=4, inflate_stages=[3], down_stages=[3])
    cfg['cls_head']['in_channels'] = 32
    np.random.seed(1)
    torch.manual_seed(1)
    m = D.build_model(cfg)
    g = torch.Generator().manual_seed(T * 10 + N)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith(('alpha', 'beta', 'add_coeff')):
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(N, 1, 2, T, 25, 3, generator=g)
    y = torch.randint(0, 12, (N, 1), generator=g)
    if kind == 'ds':
        gc = O.graph_constants('nturgb+d')
        plan = O.dgstgcn_plan(3, 16, 2, 4, (3,), (3,))
        ref_logits, ref_loss = O.recognizer_forward_train(x, y, sd, gc['node_type'], gc['edge_type'], plan)
    else:
        plan = O.ctrgcn_plan(3, 16, 4, (3,), (3,)) if kind == 'ctrgcn' else O.dgstgcn_plan(3, 16, 2, 4, (3,), (3,))
        ref_logits, ref_loss = O.recognizer_forward_train_backbone(kind, x, y, sd, plan)
    m = m.cuda().train()
    logits = m.cls_head(m.extract_feat(x.cuda()[:, 0]))
    loss = torch.nn.functional.cross_entropy(logits, y.cuda().squeeze(-1))
    loss.backward()
    assert rel(logits.detach().cpu(), ref_logits) < 1e-4
    assert abs(loss.item() - ref_loss.item()) / abs(ref_loss.item()) < 1e-4
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_gradient_packing_modes_agree():
    """FlatParams(gather=True) (autograd hands over gradient tensors, one dsgcn_pack launch packs them) must give the
    same flat gradient buffer as the in-place accumulating mode."""
    z = load('model_reduced.npz')
    with open(os.path.join(GOLD, 'model_reduced_cfg.json')) as f:
        cfg = json.load(f)
    cfg['backbone']['tcn_ms_cfg'] = [tuple(c) if isinstance(c, list) else c for c in cfg['backbone']['tcn_ms_cfg']]
    x, y = torch.from_numpy(z['x']).cuda(), torch.from_numpy(z['label']).cuda()
    flats = []
    for gather in (False, True):
        m = D.build_model(cfg)
        m.load_state_dict(sd_of(z, 'sd_', torch.float32))
        m = m.cuda().train()
        flat = D.FlatParams(m, gather=gather)
        for _ in range(2):                                   # second pass: stale gradients must not leak
            flat.zero_grad()
            m.train_step(dict(keypoint=x, label=y), None)['loss'].backward()
            flat.collect_grads()
        assert flat.check_views()
        flats.append(flat.flat_g.clone())
    assert float(flats[0].abs().max()) > 0
    assert torch.equal(flats[1], flats[0])          # every reduction on the path is ordered: bit-identical gradients


R2_CONFIGS = {
    'dsstgcn_ntu60': (lambda: ds_cfg(60), 64, 25), 'dsstgcn_ntu120': (lambda: ds_cfg(120), 64, 25),
    'dsstgcn_k400_coco': (lambda: ds_cfg(400, 'coco'), 100, 17), 'ctrgcn_ntu60': (lambda: other_cfg('ctrgcn'), 64, 25),
    'stgcn_ntu60': (lambda: other_cfg('stgcn'), 64, 25), 'stgcnpp_ntu60': (lambda: other_cfg('stgcnpp'), 64, 25),
    'ctrgcn_shipped_ntu60': (lambda: other_cfg('ctrgcn_shipped'), 64, 25),
    'stgcn_shipped_ntu60': (lambda: other_cfg('stgcn_shipped'), 64, 25)}
GRAD_CLIPS = 8


def _r2_model(name, scale=0.5):
    import sys
    sys.path.insert(0, GOLD)
    from closed_form import liven32
    mk, T, V = R2_CONFIGS[name]
    cfg = mk()
    np.random.seed(0)
    torch.manual_seed(0)
    m = D.build_model(cfg)
    liven32(m, 1, scale)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0                                       # vanilla ST-GCN's Dropout(0.5): off on both sides
    return m, cfg, T, V


@pytest.mark.parametrize('name', list(R2_CONFIGS))
def test_full_width_gradients_vs_reference_fixture(name):
    """Full-width models of every BASELINE config (1: ST-GCN, 2: DS-STGCN NTU-60, 3: NTU-120, 4: CTR-GCN, 5: K400/coco),
    ST-GCN++ and the two shipped variants, DEFAULT init + live alpha/beta/add_coeff, closed-form input of 8 clips, train
    mode: logits, loss, the FULL gradients of a fixed selection of tensors and the BatchNorm running statistics against
    what the REFERENCE produced (tests/golden/full_grads_*.npz; fp64 run = truth, its own fp32 run = yardstick)."""
    from closed_form import counter_input
    m, cfg, T, V = _r2_model(name)
    classes = cfg['cls_head']['num_classes']
    z = load(f'full_grads_{name}.npz')
    x, y = counter_input(GRAD_CLIPS, T, V, classes)
    m = m.cuda().train()
    logits = m.cls_head(m.extract_feat(x.cuda()[:, 0]))
    loss = torch.nn.functional.cross_entropy(logits, y.cuda().squeeze(-1))
    loss.backward()
    assert rel(logits.detach().cpu(), z['logits64']) < 1e-4                               # north_star bar
    assert abs(loss.item() - float(z['loss64'])) / abs(float(z['loss64'])) < 1e-4
    names = json.loads(str(z['names']))
    params = dict(m.named_parameters())
    num = den = 0.0
    per = []
    for i, k in enumerate(names):
        g64 = z[f'g64_{i}'].astype(np.float64)
        e2 = float(((params[k].grad.double().cpu().numpy() - g64) ** 2).sum())
        num += e2
        den += float((g64 ** 2).sum())
        per.append((e2, k, float((g64 ** 2).sum())))
    # The yardstick: the reference's own fp32 error on this case.  The fixture holds ONE fp32 run's error (gerr32_set) and the
    # distance between two fp32 runs of the reference whose inputs differ at the 2e-7 level (gnoise32_set): two runs with
    # independent errors e are e*sqrt(2) apart, so gnoise32_set / sqrt(2) is the error level a TYPICAL fp32 run of the
    # reference has, and the stored run may be a lucky or an unlucky draw of it.  The larger of the two is used (round 5:
    # tools/gram_check.py — the shipped CTR-GCN's ratio moves between 1.0 and 3.1 under +-1 ulp changes of ONE small tensor,
    # with a Gram kernel that is closer to fp64 than rocBLAS's; its stored run is a lucky one: 2.19e-3 against 2.77e-3).
    ours, theirs = (num / den) ** .5, max(float(z['gerr32_set']), float(z['gnoise32_set']) / 2 ** .5)
    # which tensors carry the error (share of the squared error of the selection; -s shows it: profiles/r04/parity_numbers.txt)
    per.sort(reverse=True)
    print(f'full_grads {name}: error carried by ' + ', '.join(f'{k} {e2 / max(num, 1e-300):.0%} (own rel {(e2 / max(d, 1e-300)) ** .5:.1e})'
                                                               for e2, k, d in per[:3]))
    # Whole-selection relative L2 against the reference's fp64 gradients.  ONE bar: within twice the error of the
    # reference's own fp32 run on the same case (0.05-0.8 % here: the batch-statistics backward under the mean-pooled head
    # cancels ~4 digits in ANY fp32 evaluation order, whatever the batch size — measured at 2 and 8 clips,
    # tests/golden/gen_golden_r2.py prints it).
    print(f'full_grads {name}: ours {ours:.3e}  reference fp32 {theirs:.3e} (stored run {float(z["gerr32_set"]):.3e}, run-to-run / '
          f'sqrt2 {float(z["gnoise32_set"]) / 2 ** .5:.3e})  ratio {ours / theirs:.2f}')
    assert ours < 2 * theirs, (ours, theirs)
    # The frozen mark: the ratio this configuration achieved when the marks were recorded on the GPU
    # (tests/golden/grad_ratio_marks.json, written by THIS test under DSGCN_RECORD_GRAD_RATIOS=<file> — generated, not
    # typed).  The step is bit-reproducible, so a kernel change that moves a ratio by more than 15 % — even inside the 2x
    # bar — fails here and has to be looked at.  `theirs` above is not to be redefined again.  It is a TRIPWIRE, and it
    # trips on any change of a summation order: these models amplify a last-bit difference to their fp32 noise floor
    # (profiles/r05/gram_check.txt) — splitting the K loop of the projection convs over four waves (round 6, k_pw4<.., KSP>)
    # moved NTU-120 from 0.79 to 0.95 with every per-kernel parity test unchanged (profiles/r06/grad_ratio_marks_ksp.json);
    # that form was worth 0.02 ms and stays off rather than re-recording the marks.
    ratio = ours / theirs
    rec = os.environ.get('DSGCN_RECORD_GRAD_RATIOS')
    if rec:
        marks = json.load(open(rec)) if os.path.exists(rec) else {}
        marks[name] = dict(ratio=round(ratio, 4), ours=ours, theirs=theirs)
        stamp = native.LIB_PATH + '.srchash'           # which kernel sources the marks belong to (information, not a bar)
        marks['_kernels_srchash'] = open(stamp).read().strip() if os.path.exists(stamp) else None
        with open(rec, 'w') as f:
            json.dump(marks, f, indent=1, sort_keys=True)
    else:
        with open(os.path.join(GOLD, 'grad_ratio_marks.json')) as f:
            mark = json.load(f)[name]
        assert mark['theirs'] == theirs, 'the yardstick moved: grad_ratio_marks.json was recorded against another one'
        assert ratio <= mark['ratio'] * 1.15, (name, ratio, mark['ratio'])
    sd = m.state_dict()
    for i, k in enumerate(json.loads(str(z['running_names']))):
        assert rel(sd[k].cpu(), z[f'running_{i}']) < 1e-4, k                             # F.batch_norm's running update


def _load_running(m, z):
    """BatchNorm running statistics of the eval fixture (the reference's calibrated state) into model m."""
    keys = json.loads(str(z['running_keys']))
    vals = torch.from_numpy(z['running_values'])
    sd, off = m.state_dict(), 0
    with torch.no_grad():
        for k in keys:
            n = sd[k].numel()
            sd[k].copy_(vals[off:off + n].view_as(sd[k]))
            off += n
    assert off == vals.numel()


@pytest.mark.parametrize('name', list(R2_CONFIGS))
def test_eval_mode_vs_reference_fixture(name):
    """Inference path (f-2): eval-mode BatchNorm from running statistics, 2 samples x 10 clips, scores averaged as
    probabilities (recognizergcn.py:53-107), against the REFERENCE's forward_test output and per-clip class scores."""
    from closed_form import EVAL_LIVEN, eval_clips
    m, cfg, T, V = _r2_model(name, EVAL_LIVEN.get(name, 0.5))
    z = load(f'eval_{name}.npz')
    _load_running(m, z)
    x = eval_clips(name, T, V).cuda()
    m = m.cuda().eval()
    probs = m(keypoint=x, return_loss=False)
    assert isinstance(probs, np.ndarray) and probs.shape == z['probs64'].shape
    assert rel(probs, z['probs64']) < 1e-5
    with torch.no_grad():
        scores = m.cls_head(m.extract_feat(x.flatten(0, 1))).reshape(2, 10, -1)
    assert rel(scores.cpu(), z['scores64_clips']) < 1e-4


def test_config5_batch_properties():
    """BASELINE config 5 at its per-GPU size (32 clips of 2 x 100 x 17 x 3, 400 classes): finite loss and gradients,
    clip-permutation equivariance of the logits (train-mode statistics are permutation invariant), T > 64 K-A path."""
    m, cfg, T, V = _r2_model('dsstgcn_k400_coco')
    m = m.cuda().train()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(32, 1, 2, T, V, 3, generator=g).cuda()
    y = torch.randint(0, 400, (32, 1), generator=g).cuda()
    out = m.train_step(dict(keypoint=x, label=y), None)
    out['loss'].backward()
    assert np.isfinite(out['log_vars']['loss'])
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    perm = torch.randperm(32, generator=g).cuda()
    with torch.no_grad():
        l0 = m.cls_head(m.extract_feat(x[:, 0]))
        l1 = m.cls_head(m.extract_feat(x[perm][:, 0]))
    assert rel(l1.cpu(), l0[perm].cpu()) < 1e-5


def test_full_size_properties():
    """BASELINE size (64 clips): finite loss, gradient views intact, aggregate linearity at full size."""
    from dsgcn_amd import kernels as K
    m = build_model().cuda().train()
    flat = D.FlatParams(m)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 1, 2, 64, 25, 3, generator=g).cuda()
    y = torch.randint(0, 60, (64, 1), generator=g).cuda()
    out = m.train_step(dict(keypoint=x, label=y), None)
    out['loss'].backward()
    assert np.isfinite(out['log_vars']['loss']) and flat.check_views()
    assert torch.isfinite(flat.flat_g).all()
    zp = torch.randn(128, 24, 64, 25, device='cuda')
    a1 = torch.randn(128, 24, 25, 25, device='cuda')
    a2 = torch.randn(128, 24, 25, 25, device='cuda')
    # permuting the clips permutes the logits (train-mode BN statistics are permutation-invariant): BASELINE size
    perm = torch.randperm(64, generator=g).cuda()
    with torch.no_grad():
        l0 = m.cls_head(m.extract_feat(x[:, 0]))
        l1 = m.cls_head(m.extract_feat(x[perm][:, 0]))
    assert rel(l1.cpu(), l0[perm].cpu()) < 1e-5
    # ... and leaves the parameter gradients where they were, up to the fp32 noise of a different summation order (the
    # whole-gradient error against fp64 is 4e-3 for this model: parity_numbers.txt) — the backward kernels' tiling at full
    # size (a tile form that dropped positions, as one lab experiment of round 4 did, is an O(1) error here)
    g0 = flat.flat_g.clone()
    flat.zero_grad()
    m.train_step(dict(keypoint=x[perm], label=y[perm]), None)['loss'].backward()
    assert rel(flat.flat_g.cpu(), g0.cpu()) < 2e-2, rel(flat.flat_g.cpu(), g0.cpu())
    lhs = K.aggregate(zp, None, False, a1 + a2)
    rhs = K.aggregate(zp, None, False, a1) + K.aggregate(zp, None, False, a2)
    assert rel(lhs.cpu(), rhs.cpu()) < 1e-6                   # linear in the adjacency


@pytest.mark.parametrize('kind', ['ctrgcn', 'stgcn'])
def test_full_size_properties_other_configs(kind):
    """The same properties at BASELINE size (64 clips) for config 4 (CTR-GCN: K-A', the refinement chain on 625-position
    planes, the 5-tap fused temporal stage) and config 1 (ST-GCN: K-A' with the shared adjacency, the 9-tap GEMM-form conv at
    both strides): finite loss and gradients, logits permute with the clips, and the parameter gradients stay put up to
    summation-order noise — a tile form that drops or double-counts positions at full size is an O(1) error here."""
    np.random.seed(0)
    torch.manual_seed(0)
    m = D.build_model(other_cfg(kind))
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith(('alpha', 'beta')):
                p.copy_(torch.randn(p.shape, generator=gen) * 0.5)
    m = m.cuda().train()
    flat = D.FlatParams(m)
    x = torch.randn(64, 1, 2, 64, 25, 3, generator=gen).cuda()
    y = torch.randint(0, 60, (64, 1), generator=gen).cuda()
    out = m.train_step(dict(keypoint=x, label=y), None)
    out['loss'].backward()
    assert np.isfinite(out['log_vars']['loss']) and flat.check_views() and torch.isfinite(flat.flat_g).all()
    assert float(flat.flat_g.abs().max()) > 0
    perm = torch.randperm(64, generator=gen).cuda()
    with torch.no_grad():
        l0 = m.cls_head(m.extract_feat(x[:, 0]))
        l1 = m.cls_head(m.extract_feat(x[perm][:, 0]))
    assert rel(l1.cpu(), l0[perm].cpu()) < 1e-5
    g0 = flat.flat_g.clone()
    flat.zero_grad()
    m.train_step(dict(keypoint=x[perm], label=y[perm]), None)['loss'].backward()
    assert rel(flat.flat_g.cpu(), g0.cpu()) < 2e-2, rel(flat.flat_g.cpu(), g0.cpu())


def _run_bench_child(extra_env, *args):
    """bench.py in a fresh child process (never re-exec a process that touched the GPU) -> its JSON line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **extra_env)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--no-cpu-baseline', '--no-roofline',
                          '--no-other-configs', *args],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])


def test_bench_step_under_rccl_group_is_bit_identical():
    """The N > 1 code path of bench.py / TrainEngine — graph A -> RCCL all-reduce of the flat gradient buffer -> graph B —
    under a 1-rank NCCL (= RCCL) process group (DSGCN_BENCH_FORCE_DIST=1), against the same command without a process
    group: the parameters after warm-up + 3 steps must agree bit for bit (no 8-GPU box is available to the build; this
    is the multi-GPU call sequence on real hardware)."""
    port = 29500 + os.getpid() % 400
    a = _run_bench_child({}, '--steps', '3', '--warmup', '4', '--clips-per-gpu', '8')
    b = _run_bench_child({'DSGCN_BENCH_FORCE_DIST': '1', 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port),
                          'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0'},
                         '--steps', '3', '--warmup', '4', '--clips-per-gpu', '8')
    assert a['hip_graph'] and b['hip_graph']
    assert a['param_sha256'] == b['param_sha256'], (a['param_sha256'], b['param_sha256'])
    assert a['final_loss'] == b['final_loss']
    d = b['dist']                                   # the fields a multi-GPU run is diagnosed by
    assert d['backend'] == 'nccl' and d['world_size'] == 1 and len(d['rank_ms_per_step']) == 1
    assert d['allreduce_ms'] > 0 and d['grad_bytes'] > 5_000_000 and d['rccl_version'].count('.') == 2


def test_bench_through_torch_distributed_run():
    """The driver's launch path at N > 1, on the one GPU a test box has: ``python -m torch.distributed.run --nnodes=1
    --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P bench.py --gpus 1 ...`` (reference tools/dist_train.sh:9-11) —
    the launcher is a fresh child started before anything touches the GPU, the rank reads RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* from the environment it sets, builds the RCCL group (forced at one rank) and must end bit-identical to the
    plain run."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29900 + os.getpid() % 90
    a = _run_bench_child({}, '--steps', '2', '--warmup', '4', '--clips-per-gpu', '8')
    env = dict(os.environ, DSGCN_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
                          '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(root, 'bench.py'),
                          '--gpus', '1', '--steps', '2', '--warmup', '4', '--clips-per-gpu', '8', '--no-cpu-baseline',
                          '--no-roofline', '--no-other-configs'], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    b = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert b['hip_graph'] and b['n_gpus'] == 1 and b['dist']['world_size'] == 1
    assert a['param_sha256'] == b['param_sha256'] and a['final_loss'] == b['final_loss']


@pytest.mark.parametrize('kind,switches,cache,njobs', [('ctrgcn', ('CTR_PREP_BATCH', 'CTR_FIN_BATCH'), '_ctr_prep_cache', 10),
                                                       ('stgcn', ('TSPLIT_BATCH',), '_tconv_images', 10)])
def test_engine_steps_graph_and_per_step_launches_are_bit_identical(kind, switches, cache, njobs, monkeypatch):
    """BASELINE configs 4 and 1 through TrainEngine: what depends on parameters only is built by ONE launch per step
    (kernels._StepBuilt: CTR-GCN's augmented conv4 operands — and their finishing launches, one per backward —, ST-GCN's
    per-tap weight images).  Six steps on a fixed batch as replayed hipGraphs and eagerly, with the batched launches and
    with one launch per unit: the four loss trajectories and the final parameters are the same bits (and the capture
    itself must survive the cache holding tensors across steps: a kept tensor with autograd history once took a
    default-stream dependency into the capture)."""
    from dsgcn_amd import kernels as K

    def run(graph, batch):
        for name in switches:
            monkeypatch.setattr(K, name, batch)
        getattr(K, cache).clear()
        torch.manual_seed(5)
        np.random.seed(5)
        m = D.build_model(other_cfg(kind)).cuda().train()
        eng = D.TrainEngine(m, lr=0.05, momentum=0.9, weight_decay=5e-4, nesterov=True, use_graph=graph, warmup_eager=2)
        g = torch.Generator().manual_seed(3)
        x = torch.randn(4, 1, 2, 32, 25, 3, generator=g).cuda()
        y = torch.randint(0, 60, (4, 1), generator=g).cuda()
        losses = torch.stack([eng.step(x, y)['loss'].clone() for _ in range(6)]).cpu()   # (a replay rewrites the tensor it returned)
        assert eng.graphed(x, y) == graph and torch.isfinite(losses).all()
        if batch:
            assert getattr(K, cache).batched == K._wsplit_state['epoch'] and len(getattr(K, cache).jobs) == njobs
        return losses, eng.flat.flat_p.detach().clone()
    want = run(False, False)
    for graph, batch in ((False, True), (True, False), (True, True)):
        got = run(graph, batch)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), (graph, batch, got[0], want[0])
    assert float(want[0][0]) != float(want[0][-1])


@pytest.mark.parametrize('graph', [False, True])
def test_stgcn_training_step_with_fused_dropout(graph):
    """BASELINE config 1's shipped training config (configs/stgcn/stgcn_vanilla_ntu60_xsub_3dkp/j.py:5, tcn_dropout = 0.5):
    the temporal unit's Dropout runs inside fuse_out (no materialised block, no ATen pass).  Engine steps, eager and as
    replayed hipGraphs: finite losses, the device step counter advances once per step (the replays draw new masks: the
    loss of a FIXED batch keeps changing beyond what the weight update explains is not asserted — the counter is), the same
    seed gives the same trajectory, and eval mode is untouched by dropout."""
    from dsgcn_amd import kernels as K

    def run():
        torch.manual_seed(7)
        np.random.seed(7)
        m = D.build_model(other_cfg('stgcn', tcn_dropout=0.5)).cuda().train()
        drops = [mod for mod in m.modules() if isinstance(mod, torch.nn.Dropout) and mod.p > 0]
        assert len(drops) == 9                                  # stgcn.py:105: block 0 has no dropout
        eng = D.TrainEngine(m, lr=0.05, use_graph=graph, warmup_eager=2)
        g = torch.Generator().manual_seed(3)
        x = torch.randn(4, 1, 2, 32, 25, 3, generator=g).cuda()
        y = torch.randint(0, 60, (4, 1), generator=g).cuda()
        step0 = None
        losses = []
        for i in range(6 if graph else 3):
            losses.append(eng.step(x, y)['loss'])
            cnt = int(next(iter(K._drop_state['step'].values())).item())
            step0 = cnt - 1 if step0 is None else step0
            assert cnt == step0 + i + 1
        assert eng.graphed(x, y) == graph
        losses = torch.stack(losses).cpu()
        assert torch.isfinite(losses).all()
        m.eval()
        with torch.no_grad():
            a = m.cls_head(m.extract_feat(x[:, 0]))
            b = m.cls_head(m.extract_feat(x[:, 0]))
        assert torch.equal(a, b)
        return losses, a
    (l1, a1), (l2, a2) = run(), run()
    # the device counter keeps running between the two runs, so the masks of run 2 are other draws: the trajectories differ
    # — unless the counter is put back, which is what makes a run repeatable
    assert not torch.equal(l1, l2)
    for stp in K._drop_state['step'].values():
        stp.fill_(0)
    K._drop_state['call'] = 0
    (l3, a3) = run()
    for stp in K._drop_state['step'].values():
        stp.fill_(0)
    K._drop_state['call'] = 0
    (l4, a4) = run()
    assert torch.equal(l3, l4) and torch.equal(a3, a4)


def test_bench_self_launch_path():
    """``python bench.py`` with ``DSGCN_BENCH_SELF_LAUNCH=1`` and no launcher: the parent spawns ``torch.distributed.run``
    (one rank on this box, a 1-rank RCCL group) before touching the GPU and relays rank 0's line — what
    ``python bench.py --gpus 8`` does on an 8-GPU node.  Same parameters as the plain run."""
    a = _run_bench_child({}, '--steps', '2', '--warmup', '4', '--clips-per-gpu', '8')
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DSGCN_BENCH_SELF_LAUNCH='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'DSGCN_BENCH_FORCE_DIST'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '4',
                          '--clips-per-gpu', '8', '--no-cpu-baseline', '--no-roofline', '--no-other-configs'],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'self-launch' in out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                          # ONE JSON line reaches the driver
    b = json.loads(lines[0])
    assert b['hip_graph'] and b['n_gpus'] == 1
    assert b['dist']['world_size'] == 1 and b['dist']['self_launched'] and b['dist']['backend'] == 'nccl'
    assert a['param_sha256'] == b['param_sha256'] and a['final_loss'] == b['final_loss']


@pytest.mark.parametrize('graph', [False, True])
def test_forward_after_engine_step_sees_the_updated_weights(graph):
    """the cached bf16 weight images of the wide convs (kernels._wsplit_image) must not outlive the
    optimizer update — dsgcn_sgd_step writes the flat parameter buffer through raw pointers, no version counter moves.
    Engine steps (eager, and captured + replayed), then an eval-mode forward of the engine's model against a FRESH model
    loaded with the same state_dict (other addresses: no cache entry): the same kernels on the same weights, so equal."""
    cfg = ds_cfg(60)
    torch.manual_seed(0)
    np.random.seed(0)
    m = D.build_model(cfg).cuda().train()
    eng = D.TrainEngine(m, lr=0.1, use_graph=graph, warmup_eager=2)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 1, 2, 32, 25, 3, generator=g).cuda()          # 32 frames: stages 2 and 3 take the wide-conv form
    y = torch.randint(0, 60, (4, 1), generator=g).cuda()
    for _ in range(5 if graph else 2):
        eng.step(x, y)
    assert eng.graphed(x, y) == graph
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    fresh = D.build_model(cfg).cuda()
    fresh.load_state_dict(sd)
    m.eval()
    fresh.eval()
    with torch.no_grad():
        got = m.cls_head(m.extract_feat(x[:, 0]))
        want = fresh.cls_head(fresh.extract_feat(x[:, 0]))
    assert torch.equal(got, want), rel(got.cpu(), want.cpu())
    # and against an fp64 evaluation of the CURRENT weights by the oracle
    gc = O.graph_constants('nturgb+d')
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd_cpu.items()}
    ref = O.recognizer_forward_train(x.cpu().double(), y.cpu(), sd64, gc['node_type'], gc['edge_type'], O.dgstgcn_plan(),
                                     training=False)[0]
    assert rel(got.cpu(), ref) < 1e-4, rel(got.cpu(), ref)


@pytest.mark.parametrize('name', ['model_reduced', 'model_reduced_ctrgcn', 'model_reduced_dggcn', 'model_reduced_stgcnpp'])
def test_deferred_parameter_sums_are_bit_identical(name):
    """kernels.deferred_param_sums(): the column sums of parameter-gradient partial rows queued during the backward and
    finished by one dsgcn_colsum_multi launch give exactly the gradients of the immediate sums (same ordered fp64 sums);
    backwards whose parameters reach the op through arithmetic (dggcn's expanded alpha, concatenated weights) are not
    queued and stay correct."""
    from dsgcn_amd import kernels as K
    z = load(name + '.npz')
    with open(os.path.join(GOLD, name + '_cfg.json')) as f:
        cfg = json.load(f)
    if 'tcn_ms_cfg' in cfg['backbone']:
        cfg['backbone']['tcn_ms_cfg'] = [tuple(c) if isinstance(c, list) else c for c in cfg['backbone']['tcn_ms_cfg']]
    x, y = torch.from_numpy(z['x']).cuda(), torch.from_numpy(z['label']).cuda()
    grads = []
    for deferred in (False, True):
        K.reset_leaf_uses()            # what TrainEngine does at the start of every step (leaf ids of a freed model may come back)
        m = D.build_model(cfg)
        m.load_state_dict(sd_of(z, 'sd_', torch.float32))
        m = m.cuda().train()
        flat = D.FlatParams(m, gather=True)
        flat.zero_grad()
        loss = m.train_step(dict(keypoint=x, label=y), None, sync_log_vars=False)['loss']
        if deferred:
            with K.deferred_param_sums():
                loss.backward()
                queued = len(K._deferred)
            assert queued > 0 or name == 'model_reduced_dggcn'
        else:
            loss.backward()
        flat.collect_grads()
        grads.append(flat.flat_g.clone())
    assert float(grads[0].abs().max()) > 0
    assert torch.equal(grads[0], grads[1])


@pytest.mark.parametrize('kind', ['ds', 'ctrgcn', 'stgcn', 'aagcn'])
def test_fused_step_ends_match_the_framework_ops(kind, monkeypatch):
    """The step's ends on csrc/head.hip (input BatchNorm, last block as plane means, head + loss + accuracies, BatchNorm
    buffers, SGD) against the framework's own launches for the same pieces (kernels.FUSED_ENDS = False): one TrainEngine step
    each from identical weights.  The two differ in summation order only (pooling, statistics), which ten train-mode
    BatchNorm blocks on 8 clips amplify to ~3e-5 of the loss: loss within 2e-4 (the north_star's bar is 1e-4 against fp64),
    accuracies equal, the whole gradient within 1e-2 of its norm, parameters after the step and buffers 1e-4 — a wiring
    error (a missing scale, a dropped term) is O(1) in every one of them."""
    from dsgcn_amd import kernels as K
    cfg = ds_cfg(60, 'nturgb+d') if kind == 'ds' else other_cfg(kind)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(8, 1, 2, 64, 25, 3, generator=g).cuda()
    y = torch.randint(0, 60, (8, 1), generator=g).cuda()
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(K, 'FUSED_ENDS', fused)
        np.random.seed(3)
        torch.manual_seed(3)
        m = D.build_model(cfg)
        gen = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for k, p in m.named_parameters():
                if k.endswith(('alpha', 'beta', 'add_coeff')):
                    p.copy_(torch.randn(p.shape, generator=gen) * 0.5)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        m = m.cuda().train()
        eng = D.TrainEngine(m, lr=0.05, momentum=0.9, weight_decay=5e-4, nesterov=True, use_graph=False)
        logs = eng.step(x, y)
        torch.cuda.synchronize()
        outs.append(dict(logs={k: float(v) for k, v in logs.items()}, p=eng.flat.flat_p.clone(), g=eng.flat.flat_g.clone(),
                         bufs={k: b.clone().double() for k, b in m.named_buffers()}))
    a, b = outs
    assert set(a['logs']) == set(b['logs']) == {'top1_acc', 'top5_acc', 'loss_cls', 'loss'}
    for k in a['logs']:
        bar = 2e-4 * abs(b['logs'][k]) if 'loss' in k else 0.0
        assert abs(a['logs'][k] - b['logs'][k]) <= bar, (k, a['logs'][k], b['logs'][k])
    assert float(b['g'].abs().max()) > 0
    assert rel(a['g'].cpu(), b['g'].cpu()) < 1e-2, rel(a['g'].cpu(), b['g'].cpu())
    assert rel(a['p'].cpu(), b['p'].cpu()) < 1e-4
    for k, v in b['bufs'].items():
        # (absolute floor: some running means are ~1e-11 — statistics of a sum that cancels analytically)
        diff = float((a['bufs'][k] - v).norm())
        assert diff <= 1e-4 * float(v.norm()) + 1e-6 * v.numel() ** 0.5, (k, diff, float(v.norm()))
