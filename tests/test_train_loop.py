"""The training loop (dsgcn_amd.apis.train_model: sampler order -> batches -> train_step -> backward -> all-reduce ->
cosine rate -> SGD-nesterov -> checkpoint) against a 6-step trajectory of the REFERENCE driven the way its runner drives
it (tests/golden/trajectory_dsstgcn_reduced.npz, generator tests/golden/gen_golden_r3.py: torch SGD on the imported model,
the reference's own DistributedSampler order, fp64 run = truth, its fp32 run = yardstick).

CPU: the host loop over the plain-torch op namespace (tests/torch_ops.py).  -m gpu: the product path — HIP kernels, the
step replayed from hipGraphs from the third iteration on."""
import json
import os

import numpy as np
import pytest
import torch

import dsgcn_amd as D
import torch_ops
from dsgcn_amd.apis import epoch_indices, train_model
from test_oracle_golden import GOLD, load, sd_of


def _setup(tmp_path, total_epochs=2, **extra):
    z = load('trajectory_dsstgcn_reduced.npz')
    cfg = json.loads(str(z['cfg']))
    cfg['backbone']['tcn_ms_cfg'] = [tuple(c) if isinstance(c, list) else c for c in cfg['backbone']['tcn_ms_cfg']]
    tr = json.loads(str(z['config']))
    m = D.build_model(cfg)
    m.load_state_dict(sd_of(z, 'sd_', torch.float32))
    data = [dict(keypoint=z['x'][i], label=int(z['label'][i])) for i in range(tr['samples'])]
    run_cfg = dict(data=dict(videos_per_gpu=tr['batch']), seed=tr['seed'], total_epochs=total_epochs,
                   optimizer=dict(type='SGD', lr=tr['lr'], momentum=tr['momentum'], weight_decay=tr['weight_decay'],
                                  nesterov=True),
                   optimizer_config=dict(grad_clip=None), lr_config=dict(policy='CosineAnnealing', min_lr=0, by_epoch=False),
                   checkpoint_config=dict(interval=1), log_config=dict(interval=1), work_dir=str(tmp_path), **extra)
    return z, tr, m, data, run_cfg


def _check(z, tr, m, runner, p0):
    names = json.loads(str(z['names']))
    P = dict(m.named_parameters())
    num = den = upd = 0.0
    for i, k in enumerate(names):
        want = z[f'p64_{i}']
        got = P[k].detach().double().cpu().numpy()
        num += float(((got - want) ** 2).sum())
        den += float((want ** 2).sum())
        upd += float(((want - p0[k]) ** 2).sum())
    perr, uerr = (num / den) ** .5, (num / upd) ** .5
    # the reference's own fp32 run sits 9e-7 (parameters) / 3.4e-4 (the 6-step update) from its fp64 run; dropping the
    # weight decay alone would move the update by 6e-3, plain momentum instead of nesterov or a constant rate by > 1e-1
    assert perr < 1e-5, (perr, float(z['perr32']))
    assert uerr < 3e-3, (uerr, float(z['uerr32']))
    losses = [r['loss'] for r in runner.log]
    assert len(losses) == len(z['loss64'])
    for a, b in zip(losses, z['loss64']):
        assert abs(a - b) / abs(b) < 1e-5, (losses, z['loss64'])
    lrs = [r['lr'] for r in runner.log]
    assert abs(lrs[0] - tr['lr']) < 1e-12 and abs(lrs[3] - tr['lr'] * 0.5) < 1e-12       # cosine over the 6 iterations
    sd = m.state_dict()
    for i, k in enumerate(json.loads(str(z['running_names']))):
        ref = z[f'running64_{i}'].astype(np.float64)
        got = sd[k].double().cpu().numpy()
        assert np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30) < 1e-4, k      # momentum-0.1 running statistics
    assert int(sd['backbone.data_bn.num_batches_tracked']) == int(z['nbt']) == 6
    return perr, uerr


def test_sampler_order_matches_reference():
    z = load('trajectory_dsstgcn_reduced.npz')
    tr = json.loads(str(z['config']))
    for ep in range(tr['epochs']):
        assert epoch_indices(tr['samples'], ep, tr['seed'], 0, 1) == list(z['order'][ep])
    # two ranks: the padded permutation dealt round-robin (distributed_sampler.py:38-43)
    a, b = epoch_indices(7, 0, 3, 0, 2), epoch_indices(7, 0, 3, 1, 2)
    g = torch.Generator()
    g.manual_seed(3)
    perm = torch.randperm(7, generator=g).tolist()
    perm += perm[:1]
    assert a == perm[0::2] and b == perm[1::2]


def test_train_model_trajectory_cpu(tmp_path):
    z, tr, m, data, cfg = _setup(tmp_path)
    p0 = {k: p.detach().double().numpy().copy() for k, p in m.named_parameters()}
    with D.kernels.use_ops(torch_ops):
        runner = train_model(m, data, cfg, device='cpu', use_graph=False)
    _check(z, tr, m, runner, p0)
    assert runner.epoch == 2 and runner.iter == 6
    assert os.path.islink(tmp_path / 'latest.pth') and os.path.exists(tmp_path / 'epoch_1.pth')
    final = {k: v.clone() for k, v in m.state_dict().items()}
    # a run killed after its first epoch continues from epoch_1.pth to the same end state (model, momentum, counters, rate)
    z2, _, m2, data2, cfg2 = _setup(tmp_path / 'second', resume_from=str(tmp_path / 'epoch_1.pth'))
    with D.kernels.use_ops(torch_ops):
        r2 = train_model(m2, data2, cfg2, device='cpu', use_graph=False)
    assert r2.epoch == 2 and r2.iter == 6 and len(r2.log) == 3
    for k, v in m2.state_dict().items():
        assert torch.equal(v, final[k]), k


def test_capturable_sgd_matches_torch_sgd():
    """FlatSGD(capturable=True) — rate in a device scalar, momentum buffer allocated once — steps like torch.optim.SGD."""
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
    ref = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
    ref.load_state_dict(net.state_dict())
    flat = D.FlatParams(net)
    opt = D.FlatSGD(flat, lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True, capturable=True)
    topt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True)
    assert opt.state_dict()['state'] == {}                      # nothing stepped yet: no momentum entries, like torch
    buf_ptr = opt.buf.data_ptr()
    x = torch.randn(6, 5)
    for it in range(4):
        lr = D.cosine_lr(0.1, it, 4)
        opt.set_lr(lr)
        topt.param_groups[0]['lr'] = lr
        for model, o in ((net, opt), (ref, topt)):
            o.zero_grad()
            model(x).square().sum().backward()
            o.step()
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a, b, atol=1e-6)
    # loading a checkpoint must not re-home the momentum buffer (a captured graph holds its address)
    opt.load_state_dict(topt.state_dict())
    assert opt.buf.data_ptr() == buf_ptr and abs(opt.lr - topt.param_groups[0]['lr']) < 1e-12
    assert float(opt.lr_t) == pytest.approx(opt.lr)
    off, n = flat.slices[0]
    assert torch.allclose(opt.buf[off:off + n], topt.state_dict()['state'][0]['momentum_buffer'].reshape(-1), atol=1e-6)


@pytest.mark.gpu
def test_train_model_trajectory_gpu(tmp_path):
    """The product path: HIP kernels, iterations 3-6 replayed from hipGraphs (the first two run eagerly, then the step is
    captured), lr fed through the device scalar, checkpoint + resume across the epoch boundary."""
    z, tr, m, data, cfg = _setup(tmp_path)
    p0 = {k: p.detach().double().numpy().copy() for k, p in m.named_parameters()}
    runner = train_model(m, data, cfg, device='cuda', use_graph=True)
    assert runner.engine.capture_error is None and len(runner.engine._graphs) == 1
    _check(z, tr, m, runner, p0)
    final = {k: v.clone() for k, v in m.state_dict().items()}
    z2, _, m2, data2, cfg2 = _setup(tmp_path / 'second', resume_from=str(tmp_path / 'epoch_1.pth'))
    r2 = train_model(m2, data2, cfg2, device='cuda', use_graph=True)
    assert r2.epoch == 2 and r2.iter == 6
    for k, v in m2.state_dict().items():          # every reduction on the path is ordered: eager and replayed steps agree bit for bit
        assert torch.equal(v, final[k]), k
    # eager run of the same loop == the graphed one
    z3, _, m3, data3, cfg3 = _setup(tmp_path / 'third')
    train_model(m3, data3, cfg3, device='cuda', use_graph=False)
    for k, v in m3.state_dict().items():
        assert torch.equal(v, final[k]), k


def _dp_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    z, tr, m, data, cfg = _setup(os.path.join(out_dir, f'w{rank}'), total_epochs=1)
    cfg['data']['videos_per_gpu'] = 4
    if rank == 1:                                   # a rank that starts from other weights: the wrap-time broadcast fixes it
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.01)
    with D.kernels.use_ops(torch_ops):
        runner = train_model(m, data, cfg, device='cpu', use_graph=False)
    torch.save(dict(p=runner.engine.flat.flat_p.clone(), log=runner.log, iters=runner.iter), os.path.join(out_dir, f'r{rank}.pt'))
    dist.destroy_process_group()


def test_train_model_two_ranks_gloo(tmp_path):
    """The N > 1 path of the loop on CPU: each rank walks its share of the reference sampler's order (24 clips -> 12 per
    rank, 3 iterations of 4), gradients averaged by one all-reduce per step, log scalars by one per interval: the
    replicas end bit-identical and log the same (rank-averaged) numbers."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = [torch.load(tmp_path / f'r{r}.pt', weights_only=False) for r in range(2)]
    assert r0['iters'] == r1['iters'] == 3
    assert torch.equal(r0['p'], r1['p'])
    assert [rec['loss'] for rec in r0['log']] == [rec['loss'] for rec in r1['log']]


# ---- round 4: step LR + validation during training, against the reference-generated fixture ---------------------------

def _setup_step(tmp_path, **extra):
    z = load('trajectory_stgcn_steplr.npz')
    cfg = json.loads(str(z['cfg']))
    tr = json.loads(str(z['config']))
    m = D.build_model(cfg)
    m.load_state_dict(sd_of(z, 'sd_', torch.float32))
    data = [dict(keypoint=z['x'][i], label=int(z['label'][i])) for i in range(tr['samples'])]
    val = [dict(keypoint=z['xv'][i], label=int(z['label_v'][i])) for i in range(tr['val_samples'])]
    run_cfg = dict(data=dict(videos_per_gpu=tr['batch'], val_dataloader=dict(videos_per_gpu=5)), seed=tr['seed'],
                   total_epochs=tr['epochs'],
                   optimizer=dict(type='SGD', lr=tr['lr'], momentum=tr['momentum'], weight_decay=tr['weight_decay'],
                                  nesterov=True),
                   optimizer_config=dict(grad_clip=None), lr_config=dict(policy='step', step=tr['step']),
                   evaluation=dict(interval=1, metrics=['top_k_accuracy', 'mean_class_accuracy']),
                   checkpoint_config=None, log_config=dict(interval=1), work_dir=str(tmp_path), **extra)
    return z, tr, m, data, val, run_cfg


def _check_step(z, tr, m, runner, p0, tmp_path):
    names = json.loads(str(z['names']))
    P = dict(m.named_parameters())
    num = den = upd = 0.0
    for i, k in enumerate(names):
        want = z[f'p64_{i}']
        got = P[k].detach().double().cpu().numpy()
        num += float(((got - want) ** 2).sum())
        den += float((want ** 2).sum())
        upd += float(((want - p0[k]) ** 2).sum())
    perr, uerr = (num / den) ** .5, (num / upd) ** .5
    assert perr < 1e-5 and uerr < 3e-3, (perr, uerr, float(z['perr32']), float(z['uerr32']))
    # the rate of every iteration: mmcv's StepLrUpdaterHook by epoch (0.03, then x0.1 from epoch 1, x0.01 from epoch 3)
    assert [r['lr'] for r in runner.log] == pytest.approx(list(z['lr64']), rel=1e-12)
    for a, b in zip([r['loss'] for r in runner.log], z['loss64']):
        assert abs(a - b) / abs(b) < 1e-5
    # validation after every epoch: forward_test scores (two clips averaged as probabilities) and the reference's metrics
    ev = runner.evaluator
    assert [r['epoch'] for r in ev.results] == [1, 2, 3, 4]
    for e, rec in enumerate(ev.results):
        assert rec['top1_acc'] == pytest.approx(z['val64'][e, 0]) and rec['top5_acc'] == pytest.approx(z['val64'][e, 1])
        assert rec['mean_class_accuracy'] == pytest.approx(z['val64'][e, 2])
    # save_best='auto': the first metric (top1_acc), strictly greater replaces — the fixture's top-1 never improves
    best = [f for f in os.listdir(tmp_path) if f.startswith('best_')]
    assert best == ['best_top1_acc_epoch_1.pth']
    # checkpoint_config=None: no CheckpointHook, so no epoch_N.pth / latest.pth
    assert not any(f.startswith('epoch_') or f == 'latest.pth' for f in os.listdir(tmp_path))


def test_train_model_step_lr_and_validation_cpu(tmp_path):
    z, tr, m, data, val, cfg = _setup_step(tmp_path)
    p0 = {k: p.detach().double().numpy().copy() for k, p in m.named_parameters()}
    with D.kernels.use_ops(torch_ops):
        runner = train_model(m, data, cfg, device='cpu', use_graph=False, validate=True, val_dataset=val)
        part = runner.evaluator.predict(m, 0, 1)
    _check_step(z, tr, m, runner, p0, tmp_path)
    assert np.abs(np.stack(part) - z['vscores64'][-1]).max() < 1e-5      # the last epoch's scores, sample by sample
    assert m.training                                                    # the loop hands the model back in train mode


@pytest.mark.gpu
def test_train_model_step_lr_and_validation_gpu(tmp_path):
    z, tr, m, data, val, cfg = _setup_step(tmp_path)
    p0 = {k: p.detach().double().numpy().copy() for k, p in m.named_parameters()}
    runner = train_model(m, data, cfg, device='cuda', use_graph=True, validate=True, val_dataset=val)
    assert runner.engine.capture_error is None and len(runner.engine._graphs) == 1
    _check_step(z, tr, m, runner, p0, tmp_path)
    part = runner.evaluator.predict(m, 0, 1)
    assert np.abs(np.stack(part) - z['vscores64'][-1]).max() < 1e-5


def test_log_interval_means_cpu(tmp_path):
    """log_config.interval > 1: a record is the sample-weighted mean over ITS iterations (on the hipGraph path the
    engine hands back the same static tensors every replay — the loop must keep per-iteration values)."""
    z, tr, m, data, cfg = _setup(tmp_path)
    cfg['log_config'] = dict(interval=2)
    cfg['checkpoint_config'] = None
    with D.kernels.use_ops(torch_ops):
        runner = train_model(m, data, cfg, device='cpu', use_graph=False)
    z2, _, m2, data2, cfg2 = _setup(tmp_path / 'b')
    cfg2['checkpoint_config'] = None
    with D.kernels.use_ops(torch_ops):
        r1 = train_model(m2, data2, cfg2, device='cpu', use_graph=False)
    per_iter = [r['loss'] for r in r1.log]                                # 3 iterations per epoch, 2 epochs
    want = [(per_iter[0] + per_iter[1]) / 2, per_iter[2], (per_iter[3] + per_iter[4]) / 2, per_iter[5]]
    assert [r['loss'] for r in runner.log] == pytest.approx(want, rel=1e-12)


@pytest.mark.gpu
def test_log_interval_means_graph_gpu(tmp_path):
    """The same on the product path: iterations 3-6 are hipGraph replays whose log tensors alias one static buffer."""
    z, tr, m, data, cfg = _setup(tmp_path)
    cfg['log_config'] = dict(interval=3)
    runner = train_model(m, data, cfg, device='cuda', use_graph=True)
    assert len(runner.engine._graphs) == 1
    z2, _, m2, data2, cfg2 = _setup(tmp_path / 'b')
    r1 = train_model(m2, data2, cfg2, device='cuda', use_graph=False)     # eager, one record per iteration
    per_iter = [r['loss'] for r in r1.log]
    want = [sum(per_iter[:3]) / 3, sum(per_iter[3:]) / 3]
    assert [r['loss'] for r in runner.log] == pytest.approx(want, rel=1e-6)
    top1 = [r['top1_acc'] for r in r1.log]
    assert [r['top1_acc'] for r in runner.log] == pytest.approx([sum(top1[:3]) / 3, sum(top1[3:]) / 3], rel=1e-6)


def test_eval_loop_schedule_and_best(tmp_path):
    """EvalLoop by itself: mmcv's interval / start rule and save_best bookkeeping (greater for accuracies, less for losses;
    the previous best file is removed)."""
    from dsgcn_amd.apis import EvalLoop
    ev = EvalLoop([dict(keypoint=np.zeros((1, 1, 2, 2, 3), np.float32), label=0)], interval=2, device='cpu')
    assert [e for e in range(1, 9) if ev.should_run(e)] == [2, 4, 6, 8]
    ev = EvalLoop([dict(keypoint=np.zeros((1, 1, 2, 2, 3), np.float32), label=0)], interval=2, start=3, device='cpu')
    assert [e for e in range(1, 9) if ev.should_run(e)] == [3, 5, 7]
    scores = np.array([[.1, .7, .2], [.6, .3, .1], [.2, .3, .5], [.3, .4, .3]])
    ev2 = EvalLoop([dict(keypoint=np.zeros((1, 1, 2, 2, 3), np.float32), label=0)], device='cpu',
                   metrics=['top_k_accuracy', 'mean_class_accuracy'], metric_options=dict(top_k_accuracy=dict(topk=(1, 2))))
    got = ev2.evaluate(scores, [1, 0, 1, 1])
    assert got['top1_acc'] == pytest.approx(0.75) and got['top2_acc'] == pytest.approx(1.0)
    # (the label set includes predicted classes: class 2 has no samples and counts as 0, evaluation.py:96-101)
    assert got['mean_class_accuracy'] == pytest.approx((1.0 + 2 / 3 + 0.0) / 3)
    assert ev._better('top1_acc', 0.5) and not (setattr(ev, 'best_score', 0.5) or ev._better('top1_acc', 0.5))
    assert ev._better('top1_acc', 0.6) and ev._better('loss_cls', 0.4) and not ev._better('loss_cls', 0.6)
    with pytest.raises(ValueError):
        ev._better('mystery', 1.0)


def _val_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    z, tr, m, data, val, cfg = _setup_step(os.path.join(out_dir, f'w{rank}'))
    cfg['total_epochs'] = 1
    cfg['data']['videos_per_gpu'] = 4
    val = val[:11]                                   # odd split: rank 0 scores 6 clips, rank 1 scores 5 + one wrapped
    with D.kernels.use_ops(torch_ops):
        runner = train_model(m, data, cfg, device='cpu', use_graph=False, validate=True, val_dataset=val)
        if rank == 0:
            single = runner.evaluator.predict(m, 0, 1)
            torch.save(dict(res=runner.evaluator.results, single=np.stack(single), labels=[v['label'] for v in val]),
                       os.path.join(out_dir, 'val.pt'))
    dist.destroy_process_group()


def test_validation_two_ranks_gloo(tmp_path):
    """DistEvalHook across ranks: each rank scores its share of the val split in sampler order, the parts are interleaved
    back into dataset order and cut to the dataset length; rank 0's metrics equal a single-process pass over the split."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_val_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = torch.load(tmp_path / 'val.pt', weights_only=False)
    from dsgcn_amd.evaluation import mean_class_accuracy, top_k_accuracy
    top = top_k_accuracy(r['single'], r['labels'], (1, 5))
    rec = r['res'][0]
    assert rec['epoch'] == 1 and rec['top1_acc'] == pytest.approx(top[0]) and rec['top5_acc'] == pytest.approx(top[1])
    assert rec['mean_class_accuracy'] == pytest.approx(mean_class_accuracy(r['single'], r['labels'])[0])


def _bn_sync_worker(rank, world, port, out_dir):
    import types
    import torch.distributed as dist
    from dsgcn_amd.apis import EvalLoop
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    z, tr, m, data, val, cfg = _setup_step(os.path.join(out_dir, f'w{rank}'))
    val = val[:11]
    bns = [mod for mod in m.modules() if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm)]
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for bn in bns:                                   # rank-local statistics that have drifted apart: rank 1's are far off
            bn.running_mean.copy_(torch.randn(bn.running_mean.shape, generator=g) * (0.1 if rank == 0 else 3.0))
            bn.running_var.copy_(torch.rand(bn.running_var.shape, generator=g) + (0.5 if rank == 0 else 5.0))
    out = {}
    with D.kernels.use_ops(torch_ops):
        for flag in (True, False):                       # mmcv's default first: afterwards rank 1 holds rank 0's buffers
            if not flag:
                with torch.no_grad():
                    for bn in bns:
                        if rank == 1:
                            bn.running_mean.add_(2.0)
            ev = EvalLoop(val, batch_size=5, device='cpu', metrics=['top_k_accuracy'], broadcast_bn_buffer=flag)
            runner = types.SimpleNamespace(model=m, rank=rank, world=world, epoch=1, iter=0, logger=None, work_dir=None,
                                           meta={}, engine=None)
            part = ev.predict(m, rank, world) if not flag else None
            ev.after_train_epoch(runner)
            out[flag] = dict(res=ev.results, part=None if part is None else np.stack(part),
                             stats=torch.cat([bn.running_mean for bn in bns]).clone())
        if rank == 0:
            out['single'] = np.stack(ev.predict(m, 0, 1))
            out['labels'] = [v['label'] for v in val]
    torch.save(out, os.path.join(out_dir, f'bn{rank}.pt'))
    dist.destroy_process_group()


def test_validation_broadcasts_bn_buffers_two_ranks_gloo(tmp_path):
    """mmcv's DistEvalHook broadcasts rank 0's BatchNorm running statistics before a validation pass
    (broadcast_bn_buffer=True by default); training keeps them rank-local, so without it rank 1 scores its shard with its
    own drifted statistics.  With the ranks' statistics made to differ: after the pass every rank holds rank 0's buffers and
    the gathered metrics equal a single-process pass with rank 0's model; with the option off rank 1 keeps its own."""
    import socket
    import torch.multiprocessing as mp
    from dsgcn_amd.evaluation import top_k_accuracy
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_bn_sync_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = [torch.load(tmp_path / f'bn{r}.pt', weights_only=False) for r in range(2)]
    assert torch.equal(r0[True]['stats'], r1[True]['stats'])
    top = top_k_accuracy(r0['single'], r0['labels'], (1, 5))
    assert r0[True]['res'][0]['top1_acc'] == pytest.approx(top[0]) and r0[True]['res'][0]['top5_acc'] == pytest.approx(top[1])
    assert not torch.equal(r0[False]['stats'], r1[False]['stats'])              # off: rank-local statistics stay
    # ... and rank 1's shard was then scored with ITS statistics: its scores differ from rank 0's model on the same clips
    assert np.abs(r1[False]['part'][:5] - r0['single'][1::2]).max() > 1e-3       # (its 6th entry is the wrapped sample 0)


def test_resume_restores_the_best_score(tmp_path):
    """the best score / file so far ride in every epoch checkpoint's meta['hook_msgs'] (mmcv EvalHook), so the
    first evaluation after a resume competes with them instead of always counting as 'better'.  The fixture's top-1 never
    improves after epoch 1: a run resumed from epoch_2.pth must leave best_top1_acc_epoch_1.pth the only best file."""
    z, tr, m, data, val, cfg = _setup_step(tmp_path)
    cfg.update(total_epochs=2, checkpoint_config=dict(interval=1))
    with D.kernels.use_ops(torch_ops):
        train_model(m, data, cfg, device='cpu', use_graph=False, validate=True, val_dataset=val)
    meta = torch.load(tmp_path / 'epoch_2.pth', weights_only=False)['meta']
    assert meta['hook_msgs']['best_ckpt'].endswith('best_top1_acc_epoch_1.pth') and meta['hook_msgs']['key_indicator'] == 'top1_acc'
    z, tr, m2, data, val, cfg = _setup_step(tmp_path)
    cfg.update(total_epochs=3, checkpoint_config=dict(interval=1), resume_from=str(tmp_path / 'epoch_2.pth'))
    with D.kernels.use_ops(torch_ops):
        runner = train_model(m2, data, cfg, device='cpu', use_graph=False, validate=True, val_dataset=val)
    assert [r['epoch'] for r in runner.evaluator.results] == [3]
    assert runner.evaluator.best_score == meta['hook_msgs']['best_score']
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith('best_')) == ['best_top1_acc_epoch_1.pth']


def test_validation_leaves_the_training_rng_stream_alone():
    """the test-mode sampler reseeds numpy's global RNG for every val clip; run inside the training process that
    must not move the training augmentation stream (the reference samples val clips in separate loader workers)."""
    from dsgcn_amd.apis import EvalLoop
    from dsgcn_amd import pipeline as P

    class _Val:
        def __len__(self):
            return 3

        def __getitem__(self, i):
            P.uniform_frame_indices(20, 8, 2, test_mode=True)                 # np.random.seed(255) inside
            return dict(keypoint=np.zeros((2, 2, 8, 25, 3), np.float32), label=0)

    class _Model(torch.nn.Module):
        def forward(self, keypoint, return_loss=False):
            return [np.ones(4) / 4] * len(keypoint)

    np.random.seed(11)
    want = np.random.rand(3)
    np.random.seed(11)
    EvalLoop(_Val(), batch_size=2, device='cpu').predict(_Model(), 0, 1)
    assert np.array_equal(np.random.rand(3), want)


@pytest.mark.gpu
def test_train_model_on_resident_store_feeder_is_deterministic(tmp_path):
    """train_model fed by the HIP input pipeline (clips resident in HBM, host plan + one launch per batch): with the plan
    made one batch ahead on the feeder thread or inline, the same numpy RNG stream is consumed — bit-identical weights."""
    import sys
    sys.path.insert(0, GOLD)
    from pipeline_cases import annotations, pipelines
    from dsgcn_amd import pipeline as P
    anns = [dict(a, label=a['label'] % 12) for a in annotations() * 6]      # 18+ clips, the reduced model's 12 classes
    cfg_m = json.loads(str(load('trajectory_dsstgcn_reduced.npz')['cfg']))
    cfg_m['backbone']['tcn_ms_cfg'] = [tuple(c) if isinstance(c, list) else c for c in cfg_m['backbone']['tcn_ms_cfg']]
    name = 'train_j'
    finals = []
    for prefetch in (True, False):
        torch.manual_seed(0)
        np.random.seed(0)
        m = D.build_model(cfg_m)
        store = P.SkeletonStore(anns)
        batcher = P.SkeletonBatcher(pipelines()[name])
        run_cfg = dict(data=dict(videos_per_gpu=4), seed=1, total_epochs=2,
                       optimizer=dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=5e-4, nesterov=True),
                       lr_config=dict(policy='CosineAnnealing', min_lr=0, by_epoch=False), checkpoint_config=None,
                       log_config=dict(interval=2), work_dir=None)
        np.random.seed(5)
        r = train_model(m, (store, batcher), run_cfg, device='cuda', use_graph=True, prefetch=prefetch)
        assert r.iter == 2 * ((len(anns) + 3) // 4) and all(np.isfinite(rec['loss']) for rec in r.log)
        finals.append({k: v.clone() for k, v in m.state_dict().items()})
    for k, v in finals[0].items():
        assert torch.equal(v, finals[1][k]), k
