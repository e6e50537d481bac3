"""CPU: the oracle (oracle/dsgcn_oracle.py) against the committed golden vectors (tests/golden/*.npz, generated from the
imported reference by tests/golden/gen_golden.py) and, where /root/reference exists, against the reference itself."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import dsgcn_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return dict(np.load(os.path.join(GOLD, name)))


def sd_of(npz, prefix, dtype=torch.float64):
    out = {}
    for k, v in npz.items():
        if k.startswith(prefix):
            t = torch.from_numpy(v)
            out[k[len(prefix):]] = t.to(dtype) if t.dtype.is_floating_point else t
    return out


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)


def test_graph_constants():
    g = load('graph_constants.npz')
    for lay, tag in (('nturgb+d', 'nturgbpd'), ('coco', 'coco')):
        c = O.graph_constants(lay)
        assert np.array_equal(c['node_type'], g[f'{tag}_node_type'])
        assert np.array_equal(c['edge_type'], g[f'{tag}_edge_type'])          # integer tables: bit-exact
        assert np.array_equal(O.graph_A(lay, 'spatial'), g[f'{tag}_spatial'])
        assert np.array_equal(O.graph_A(lay, 'stgcn_spatial'), g[f'{tag}_stgcn_spatial'])


@pytest.mark.parametrize('i', [0, 1, 2])
def test_dgphgcn1_unit(i):
    z = load('unit_dgphgcn1.npz')
    tag = f'u{i}_'
    sd = sd_of(z, tag + 'sd_')
    gc = O.graph_constants('nturgb+d')
    x = torch.from_numpy(z[tag + 'x']).double().requires_grad_()
    for k in ('A', 'alpha', 'beta', 'edge_linears.weight', 'conv1_se.weight', 'pre.1.weight'):
        sd[k].requires_grad_()
    y = O.dgphgcn1_forward(x, sd, gc['node_type'], gc['edge_type'])
    (y * torch.from_numpy(z[tag + 'R']).double()).sum().backward()
    # goldens are fp64 results stored as fp32: 1e-6 relative
    assert rel(y.detach(), z[tag + 'y']) < 1e-6
    assert rel(x.grad, z[tag + 'dx']) < 1e-6
    for k in ('A', 'alpha', 'beta', 'edge_linears.weight', 'conv1_se.weight', 'pre.1.weight'):
        assert rel(sd[k].grad, z[tag + 'grad_' + k]) < 1e-6, k


DGGCN_GRADS = ('A', 'alpha', 'beta', 'conv1.weight', 'conv2.bias', 'pre.1.weight', 'post.weight')


@pytest.mark.parametrize('i', [0, 1])
def test_dggcn_unit(i):
    """The original DG-STGCN unit (gcn.py:1445-1584; f-4): scalar and subset-wise alpha / beta, with and without `down`."""
    z = load('unit_dggcn.npz')
    tag = f'u{i}_'
    sd = sd_of(z, tag + 'sd_')
    x = torch.from_numpy(z[tag + 'x']).double().requires_grad_()
    for k in DGGCN_GRADS:
        sd[k].requires_grad_()
    y = O.dggcn_forward(x, sd, subset_wise=bool(z[tag + 'subset_wise']))
    (y * torch.from_numpy(z[tag + 'R']).double()).sum().backward()
    assert rel(y.detach(), z[tag + 'y']) < 1e-6
    assert rel(x.grad, z[tag + 'dx']) < 1e-6
    for k in DGGCN_GRADS:
        want = z[tag + 'grad_' + k]
        if np.abs(want).max() < 1e-12:           # alpha[1:], beta[1:] are unused without subset_wise
            assert sd[k].grad is None or float(sd[k].grad.abs().max()) < 1e-12, k
        else:
            assert rel(sd[k].grad, want) < 1e-6, k


AAGCN_GRADS = ('A', 'alpha', 'conv_d.1.weight', 'conv_a.0.weight', 'conv_b.2.bias', 'conv_ta.weight', 'conv_sa.weight',
               'fc1c.weight', 'fc2c.bias', 'bn.weight')


@pytest.mark.parametrize('i', [0, 1])
def test_unit_aagcn(i):
    """2s-AGCN / AAGCN unit (gcn.py:349-460; f-4): adaptive topology + the three attention gates, with and without `down`."""
    z = load('unit_aagcn.npz')
    tag = f'u{i}_'
    sd = sd_of(z, tag + 'sd_')
    x = torch.from_numpy(z[tag + 'x']).double().requires_grad_()
    for k in AAGCN_GRADS:
        sd[k].requires_grad_()
    y = O.unit_aagcn_forward(x, sd)
    (y * torch.from_numpy(z[tag + 'R']).double()).sum().backward()
    assert rel(y.detach(), z[tag + 'y']) < 1e-6
    assert rel(x.grad, z[tag + 'dx']) < 1e-6
    for k in AAGCN_GRADS:
        assert rel(sd[k].grad, z[tag + 'grad_' + k]) < 1e-6, k


def test_unit_aagcn_vs_reference_live():
    import ref_shim
    if not ref_shim.available():
        pytest.skip('reference tree not present (GPU box)')
    R = ref_shim.load()
    torch.manual_seed(4)
    A = torch.tensor(R.graph.Graph(layout='coco', mode='spatial').A, dtype=torch.float32)
    for ci, co in ((48, 48), (32, 64)):
        m = R.gutils.unit_aagcn(ci, co, A.clone())
        m.init_weights()
        with torch.no_grad():
            m.alpha.normal_(0, .5)
            m.conv_ta.weight.normal_(0, .1)
            m.fc2c.weight.normal_(0, .05)
            m.bn.weight.fill_(1.0)
        m = m.double()
        x = torch.randn(2, ci, 10, 17, dtype=torch.float64)
        sd = {k: v.detach() for k, v in m.state_dict().items()}
        assert (m(x) - O.unit_aagcn_forward(x, sd)).abs().max().item() < 1e-11


def test_reduced_dggcn_model():
    """DGSTGCN(gcn_type='dggcn') end to end (reduced width) against the reference's fp64 and fp32 runs."""
    z = load('model_reduced_dggcn.npz')
    with open(os.path.join(GOLD, 'model_reduced_dggcn_cfg.json')) as f:
        bk = json.load(f)['backbone']
    plan = O.dgstgcn_plan(3, bk['base_channels'], 2, bk['num_stages'], tuple(bk['inflate_stages']), tuple(bk['down_stages']))
    gc = O.graph_constants('nturgb+d')
    x, y = torch.from_numpy(z['x']), torch.from_numpy(z['label'])
    logits, loss = O.recognizer_forward_train(x.double(), y, sd_of(z, 'sd_'), gc['node_type'], gc['edge_type'], plan)
    assert rel(logits, z['logits_f64']) < 1e-6
    assert abs(loss.item() - float(z['loss_f64'])) < 1e-7
    logits32, loss32 = O.recognizer_forward_train(x, y, sd_of(z, 'sd_', torch.float32), gc['node_type'], gc['edge_type'], plan)
    assert rel(logits32, z['logits_f32']) < 1e-5
    assert abs(loss32.item() - float(z['loss_f32'])) < 1e-5


@pytest.mark.parametrize('i,stride', [(0, 1), (1, 2)])
def test_dgmstcn_unit(i, stride):
    z = load('unit_dgmstcn.npz')
    tag = f't{i}_'
    sd = sd_of(z, tag + 'sd_')
    sd['add_coeff'].requires_grad_()
    x = torch.from_numpy(z[tag + 'x']).double().requires_grad_()
    y = O.dgmstcn_forward(x, sd, stride)
    (y * torch.from_numpy(z[tag + 'R']).double()).sum().backward()
    assert rel(y.detach(), z[tag + 'y']) < 1e-6
    assert rel(x.grad, z[tag + 'dx']) < 1e-6
    assert rel(sd['add_coeff'].grad, z[tag + 'grad_add_coeff']) < 1e-6


def _reduced():
    z = load('model_reduced.npz')
    with open(os.path.join(GOLD, 'model_reduced_cfg.json')) as f:
        cfg = json.load(f)
    bk = cfg['backbone']
    plan = O.dgstgcn_plan(3, bk['base_channels'], 2, bk['num_stages'], tuple(bk['inflate_stages']),
                          tuple(bk['down_stages']))
    return z, cfg, plan


def test_reduced_model_fp64_and_fp32():
    z, cfg, plan = _reduced()
    gc = O.graph_constants('nturgb+d')
    x = torch.from_numpy(z['x'])
    y = torch.from_numpy(z['label'])
    sd64 = sd_of(z, 'sd_')
    logits, loss = O.recognizer_forward_train(x.double(), y, sd64, gc['node_type'], gc['edge_type'], plan)
    assert rel(logits, z['logits_f64']) < 1e-6
    assert abs(loss.item() - float(z['loss_f64'])) < 1e-9 * max(1, abs(float(z['loss_f64']))) + 1e-7
    sd32 = sd_of(z, 'sd_', torch.float32)
    logits32, loss32 = O.recognizer_forward_train(x, y, sd32, gc['node_type'], gc['edge_type'], plan)
    # fp32 oracle vs the reference's own fp32 output: same op sequence, 1e-5 relative
    assert rel(logits32, z['logits_f32']) < 1e-5
    assert abs(loss32.item() - float(z['loss_f32'])) < 1e-5


@pytest.mark.parametrize('kind', ['ctrgcn', 'stgcn', 'stgcnpp', 'aagcn', 'stgcn_shipped'])
def test_reduced_other_backbones(kind):
    """ST-GCN (unit_gcn + unit_tcn k=9), ST-GCN++ (with_res + mstcn), classic CTR-GCN (unit_ctrgcn + MSTCN) and AAGCN
    (unit_aagcn + unit_tcn k=9, data_bn over M V C) end to end against the reference."""
    z = load(f'model_reduced_{kind}.npz')
    with open(os.path.join(GOLD, f'model_reduced_{kind}_cfg.json')) as f:
        bk = json.load(f)['backbone']
    if kind == 'ctrgcn':
        plan = O.ctrgcn_plan(3, bk['base_channels'], bk['num_stages'], tuple(bk['inflate_stages']), tuple(bk['down_stages']))
    elif kind == 'aagcn':
        plan = O.aagcn_plan(3, bk['base_channels'], bk['num_stages'], tuple(bk['inflate_stages']), tuple(bk['down_stages']))
    else:
        plan = O.dgstgcn_plan(3, bk['base_channels'], 2, bk['num_stages'], tuple(bk['inflate_stages']),
                              tuple(bk['down_stages']))
    x, y = torch.from_numpy(z['x']), torch.from_numpy(z['label'])
    logits, loss = O.recognizer_forward_train_backbone(kind, x.double(), y, sd_of(z, 'sd_'), plan)
    assert rel(logits, z['logits_f64']) < 1e-6
    assert abs(loss.item() - float(z['loss_f64'])) < 1e-7
    logits32, loss32 = O.recognizer_forward_train_backbone(kind, x, y, sd_of(z, 'sd_', torch.float32), plan)
    assert rel(logits32, z['logits_f32']) < 1e-5
    assert abs(loss32.item() - float(z['loss_f32'])) < 1e-5


def test_full_size_closed_form_ds():
    """Full-width DS-STGCN (BASELINE config 2 shapes, 2 clips) with closed-form weights: oracle vs the reference's
    stored logits / loss.  Both sides are fp32 evaluations of a 10-block network with different reduction orders
    (measured 1.8e-5 apart): the bar is the north_star's 1e-4."""
    import sys
    sys.path.insert(0, GOLD)
    from closed_form import closed_form_fill, counter_input
    import dsgcn_amd as D
    from bench import ds_cfg
    np.random.seed(0)
    torch.manual_seed(0)
    m = D.build_model(ds_cfg(60, 'nturgb+d'))
    closed_form_fill(m)
    z = load('full_size.npz')
    x, y = counter_input(2, 64, 25, 60)
    gc = O.graph_constants('nturgb+d')
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        logits, loss = O.recognizer_forward_train(x, y, sd, gc['node_type'], gc['edge_type'], O.dgstgcn_plan())
    assert rel(logits, z['dsstgcn_ntu60_logits']) < 1e-4
    assert abs(loss.item() - float(z['dsstgcn_ntu60_loss'])) < 1e-4


def test_oracle_vs_reference_live():
    import ref_shim
    if not ref_shim.available():
        pytest.skip('reference tree not present (GPU box)')
    R = ref_shim.load()
    torch.manual_seed(1)
    np.random.seed(1)
    G = R.graph.Graph(layout='coco', mode='random', num_filter=3, init_off=.04, init_std=.02)
    A = torch.tensor(G.A, dtype=torch.float32)
    m = R.gutils.dgphgcn1(64, 64, A, torch.tensor(G.edge_type, dtype=torch.float32), torch.tensor(G.node_type),
                          ratio=0.125, decompose=True, node_attention=True, edge_attention=True, subset_wise=True,
                          ctr='T', ada='T').double()
    with torch.no_grad():
        m.alpha.normal_(0, .5)
        m.beta.normal_(0, .5)
    x = torch.randn(2, 64, 10, 17, dtype=torch.float64)
    gc = O.graph_constants('coco')
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    assert (m(x) - O.dgphgcn1_forward(x, sd, gc['node_type'], gc['edge_type'])).abs().max().item() < 1e-12


def test_dggcn_vs_reference_live():
    """dggcn restatement against the imported reference class, fp64, other widths / joints than the fixture's."""
    import ref_shim
    if not ref_shim.available():
        pytest.skip('reference tree not present (GPU box)')
    R = ref_shim.load()
    torch.manual_seed(2)
    np.random.seed(2)
    G = R.graph.Graph(layout='coco', mode='random', num_filter=3, init_off=.04, init_std=.02)
    A = torch.tensor(G.A, dtype=torch.float32)
    for ci, co, sw in ((48, 48, False), (32, 64, True)):
        m = R.gutils.dggcn(ci, co, A, ratio=0.25, subset_wise=sw).double()
        with torch.no_grad():
            m.alpha.normal_(0, .5)
            m.beta.normal_(0, .5)
        x = torch.randn(2, ci, 10, 17, dtype=torch.float64)
        sd = {k: v.detach() for k, v in m.state_dict().items()}
        assert (m(x) - O.dggcn_forward(x, sd, subset_wise=sw)).abs().max().item() < 1e-12


# ---- round-2 fixtures (tests/golden/gen_golden_r2.py) ------------------------------------------------------------

@pytest.mark.parametrize('tag,layout', [('v25', 'nturgb+d'), ('v17', 'coco')])
def test_dgphgcn1_intermediates(tag, layout):
    """Every intermediate of the reference's dgphgcn1 forward (captured from its own einsum / conv calls): xbar, x1,
    x2, tanh(D), softmax(G), Ahat, P, Y, output — the oracle reproduces each one, not only the unit's output."""
    z = load('unit_intermediates.npz')
    sd = sd_of({k[len(tag) + 1:]: v for k, v in z.items() if k.startswith(tag + '_')}, 'sd_')
    x = torch.from_numpy(z[tag + '_x']).double()
    gc = O.graph_constants(layout)
    assert np.array_equal(np.asarray(gc['node_type']), z[tag + '_node_type'])
    assert np.array_equal(np.asarray(gc['edge_type']), z[tag + '_edge_type'])
    y, p = O.dgphgcn1_forward(x, sd, gc['node_type'], gc['edge_type'], ret_parts=True)
    got = dict(xbar=p['xbar'], x1=p['x1'], x2=p['x2'], tanhD=torch.tanh(p['D']), softG=p['Sm'], Ahat=p['Ahat'], P=p['P'],
               Y=p['Y'], out=y)
    for k, v in got.items():
        assert rel(v, z[f'{tag}_{k}']) < 2e-6, (k, rel(v, z[f'{tag}_{k}']))      # fixtures are stored as fp32


def _graph_tables():
    gc = O.graph_constants('nturgb+d')
    return np.asarray(gc['edge_type']), np.asarray(gc['node_type'])


def _unit_oracle(tag, x, sd):
    if tag.startswith('ctrhgcn'):
        return O.unit_ctrhgcn_forward(x, sd, _graph_tables()[0])
    if tag == 'msmlp':
        return O.msmlp_forward(x, sd, 1, merge_after=True)
    if tag == 'msmlp_s2':
        return O.msmlp_forward(x, sd, 2, merge_after=False)
    if tag in ('gcn', 'gcn_res'):
        return O.unit_gcn_forward(x, sd, with_res=(tag == 'gcn_res'))
    if tag == 'gcn_offset_post':
        return O.unit_gcn_forward(x, sd, adaptive='offset', conv_pos='post')
    if tag == 'gcn_importance':
        return O.unit_gcn_forward(x, sd, with_res=True, adaptive='importance')
    if tag == 'gcn_fixed_post':
        return O.unit_gcn_forward(x, sd, adaptive=None, conv_pos='post')
    if tag == 'tcn9':
        return O.unit_tcn_forward(x, sd, 9, 1)
    if tag == 'tcn1s2':
        return O.unit_tcn_forward(x, sd, 1, 2)
    if tag == 'ctrgcn':
        return O.unit_ctrgcn_forward(x, sd)
    if tag == 'MSTCN':
        return O.mstcn_msg3d_forward(x, sd, 1, 5, (1, 2))
    if tag in ('unitmlp9', 'unitmlp9_s2'):        # unitmlp as a whole unit: its own BatchNorm closes it (tcn.py:609)
        s2 = tag.endswith('_s2')
        return O._bn(O.unitmlp_forward(x, sd, 9, 2 if s2 else 1, 1, True, not s2), sd, 'bn.', True)
    raise KeyError(tag)


@pytest.mark.parametrize('tag', ['gcn', 'gcn_res', 'tcn9', 'tcn1s2', 'ctrgcn', 'MSTCN', 'gcn_offset_post', 'gcn_importance',
                                 'gcn_fixed_post', 'ctrhgcn', 'ctrhgcn_same', 'msmlp', 'msmlp_s2', 'unitmlp9', 'unitmlp9_s2'])
def test_other_units_vs_reference_fixture(tag):
    """unit_gcn, unit_tcn (k=9; k=1 stride 2), unit_ctrgcn, MSTCN at real widths: oracle output and input gradient
    against the reference's (weights rebuilt from the shared seeded recipe; their digest is part of the fixture)."""
    import sys
    sys.path.insert(0, GOLD)
    from closed_form import make_unit, sd_digest
    import dsgcn_amd as D
    z = load('unit_others.npz')
    A = torch.tensor(O.graph_A('nturgb+d', 'spatial'), dtype=torch.float32)
    m, x, Rm = make_unit(D, tag, A, *_graph_tables())
    assert sd_digest(m) == str(z[f'{tag}_digest']), 'seeded unit weights differ from the reference build'
    sd = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in m.state_dict().items()}
    x = x.double().requires_grad_()
    y = _unit_oracle(tag, x, sd)
    (y * Rm.double()).sum().backward()
    assert rel(y.detach(), z[f'{tag}_y']) < 2e-6
    assert rel(x.grad, z[f'{tag}_dx']) < 2e-6


@pytest.mark.parametrize('name,kind,T,V,classes,layout', [
    ('dsstgcn_ntu60', 'ds', 64, 25, 60, 'nturgb+d'), ('stgcnpp_ntu60', 'stgcnpp', 64, 25, 60, 'nturgb+d'),
    ('ctrgcn_shipped_ntu60', 'ctrgcn_shipped', 64, 25, 60, 'nturgb+d')])
def test_eval_fixture_oracle(name, kind, T, V, classes, layout):
    """Eval-mode (running statistics) test-time scores of the reference, 2 samples x 10 clips averaged as
    probabilities (recognizergcn.py:53-107): the oracle's inference path against the stored fp64 scores."""
    import sys
    sys.path.insert(0, GOLD)
    from closed_form import EVAL_LIVEN, eval_clips, liven32
    import dsgcn_amd as D
    from bench import ds_cfg, other_cfg
    np.random.seed(0)
    torch.manual_seed(0)
    m = D.build_model(ds_cfg(classes, layout) if kind == 'ds' else other_cfg(kind))
    liven32(m, 1, EVAL_LIVEN.get(name, 0.5))
    z = load(f'eval_{name}.npz')
    keys, vals, off = json.loads(str(z['running_keys'])), torch.from_numpy(z['running_values']), 0
    with torch.no_grad():
        for k in keys:                                # the reference's calibrated running statistics
            t = m.state_dict()[k]
            t.copy_(vals[off:off + t.numel()].view_as(t))
            off += t.numel()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = eval_clips(name, T, V)[:, :2]              # 2 of the 10 clips keep the CPU suite fast (-m gpu checks all 10)
    gc = O.graph_constants(layout)
    flat = x.flatten(0, 1)
    label = torch.zeros(flat.shape[0], 1, dtype=torch.long)
    with torch.no_grad():
        if kind == 'ds':
            logits, _ = O.recognizer_forward_train(flat[:, None], label, sd, gc['node_type'], gc['edge_type'],
                                                   O.dgstgcn_plan(), training=False)
        else:
            plan = O.ctrgcn_plan() if kind.startswith('ctrgcn') else O.dgstgcn_plan()
            logits, _ = O.recognizer_forward_train_backbone(kind, flat[:, None], label, sd, plan, training=False)
    assert rel(logits.reshape(2, 2, -1), z['scores64_clips'][:, :2]) < 1e-4          # class scores of each clip
    assert rel(torch.softmax(logits.reshape(2, 2, -1), 2), z['probs64_clips'][:, :2]) < 1e-5


@pytest.mark.parametrize('tag', ['gcn', 'gcn_res', 'tcn9', 'tcn1s2', 'ctrgcn', 'MSTCN', 'gcn_offset_post', 'gcn_importance',
                                 'gcn_fixed_post', 'ctrhgcn', 'ctrhgcn_same', 'msmlp', 'msmlp_s2', 'unitmlp9', 'unitmlp9_s2'])
def test_oracle_units_vs_reference_live(tag):
    """Build container only: every unit restated in the oracle against the IMPORTED reference module, fp64."""
    import sys
    sys.path.insert(0, GOLD)
    import ref_shim
    if not ref_shim.available():
        pytest.skip('reference tree not present (GPU box)')
    from closed_form import make_unit
    R = ref_shim.load()
    Gs = R.graph.Graph(layout='nturgb+d', mode='spatial')
    A = torch.tensor(Gs.A, dtype=torch.float32)
    m, x, _ = make_unit(R.gutils, tag, A, Gs.edge_type, Gs.node_type)
    m = m.double().train()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    want = m(x.double())
    got = _unit_oracle(tag, x.double(), sd)
    assert (want - got).abs().max().item() < 1e-11


@pytest.mark.parametrize('cls,stride', [('dgmstcn', 1), ('dgmstcn', 2), ('mstcn', 1), ('mstcn', 2)])
def test_oracle_temporal_units_vs_reference_live(cls, stride):
    import sys
    sys.path.insert(0, GOLD)
    import ref_shim
    if not ref_shim.available():
        pytest.skip('reference tree not present (GPU box)')
    R = ref_shim.load()
    torch.manual_seed(5)
    m = getattr(R.gutils, cls)(64, 64, stride=stride).double().train()
    with torch.no_grad():
        if hasattr(m, 'add_coeff'):
            m.add_coeff.normal_(0, .5)
    x = torch.randn(2, 64, 10, 25, dtype=torch.float64)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    want = m(x)
    got = (O.dgmstcn_forward if cls == 'dgmstcn' else O.mstcn_forward)(x, sd, stride)
    assert (want - got).abs().max().item() < 1e-11


# ---- round-3 fixtures (tests/golden/gen_golden_r3.py) ------------------------------------------------------------

def _ds_unit_tags():
    import sys
    sys.path.insert(0, GOLD)
    from closed_form import DS_UNIT_CASES
    return list(DS_UNIT_CASES)


def _ds_unit_oracle(tag, x, sd):
    import sys
    sys.path.insert(0, GOLD)
    from closed_form import DS_UNIT_CASES
    cls, layout, kw, _ = DS_UNIT_CASES[tag]
    if cls == 'dgphgcn1':
        gc = O.graph_constants(layout)
        return O.dgphgcn1_forward(x, sd, gc['node_type'], gc['edge_type'])
    return O.dgmstcn_forward(x, sd, kw.get('stride', 1))


@pytest.mark.parametrize('tag', _ds_unit_tags())
def test_ds_units_all_widths_oracle(tag):
    """dgphgcn1 / dgmstcn at every width DS-STGCN uses: oracle output, input gradient and EVERY parameter gradient against
    the reference's fp64 run (tests/golden/unit_ds_r3.npz; weights rebuilt from the shared seeded recipe, digest checked)."""
    import dsgcn_amd as D
    from closed_form import make_ds_unit, sd_digest
    z = load('unit_ds_r3.npz')
    m, x, Rm = make_ds_unit(D, D.Graph, tag)
    assert sd_digest(m) == str(z[f'{tag}_digest']), 'seeded unit weights differ from the reference build'
    sd = {k: (v.detach().double().requires_grad_() if v.dtype.is_floating_point and 'running' not in k else v.detach().clone())
          for k, v in m.state_dict(keep_vars=True).items()}
    x = x.double().requires_grad_()
    y = _ds_unit_oracle(tag, x, sd)
    (y * Rm.double()).sum().backward()
    assert rel(y.detach(), z[f'{tag}_y']) < 2e-6
    assert rel(x.grad, z[f'{tag}_dx']) < 2e-6
    checked = 0
    for k, v in sd.items():
        if f'{tag}_grad_{k}' in z:
            want = z[f'{tag}_grad_{k}']
            if np.abs(want).max() < 1e-9:
                assert v.grad is None or float(v.grad.abs().max()) < 1e-9, k
            else:
                assert rel(v.grad, want) < 2e-6, k
            checked += 1
        elif f'{tag}_gnorm_{k}' in z:
            assert abs(float(v.grad.norm()) - float(z[f'{tag}_gnorm_{k}'])) < 2e-6 * float(z[f'{tag}_gnorm_{k}']), k
            checked += 1
    assert checked >= 17
