"""CPU, 2 processes, gloo: the flat-buffer data-parallel path == one process evaluating the two rank batches as
independent micro-batches (rank-local BatchNorm) and averaging gradients (SURVEY §8e equivalence check)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import dsgcn_amd as D
import torch_ops

CFG = dict(type='RecognizerGCN',
           backbone=dict(type='DGSTGCN', gcn_type='dgphgcn1', gcn_ratio=0.125, gcn_node_attention=True,
                         gcn_edge_attention=True, gcn_decompose=True, gcn_subset_wise=True, gcn_ctr='T', gcn_ada='T',
                         tcn_type='dgmstcn', base_channels=16, num_stages=3, inflate_stages=[3], down_stages=[3],
                         graph_cfg=dict(layout='nturgb+d', mode='random', num_filter=3, init_off=.04, init_std=.02),
                         tcn_ms_cfg=[(3, 1), (3, 2), (3, 3), (3, 4), ('max', 3), '1x1']),
           cls_head=dict(type='GCNHead', num_classes=7, in_channels=32))


def make_model(seed):
    np.random.seed(seed)
    torch.manual_seed(seed)
    m = D.build_model(CFG)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith(('alpha', 'beta', 'add_coeff')):
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)
    return m.train()


def batch_for(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return dict(keypoint=torch.randn(2, 1, 2, 8, 25, 3, generator=g), label=torch.randint(0, 7, (2, 1), generator=g))


def _worker(rank, world, port, out_dir, gather):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = make_model(seed=7 + rank)           # different init per rank: the param broadcast must fix it
    with torch.no_grad():                       # ... and different buffers (a rank that loaded another checkpoint)
        model.backbone.data_bn.running_mean.fill_(float(rank))
        model.backbone.data_bn.num_batches_tracked.fill_(10 * rank)
    flat = D.FlatParams(model, gather=gather)
    dp = D.FlatDataParallel(flat)
    # wrap time: rank 0's parameters AND buffers everywhere (torch DDP's constructor syncs module states incl. buffers)
    assert float(model.backbone.data_bn.running_mean.abs().max()) == 0.0
    assert int(model.backbone.data_bn.num_batches_tracked) == 0
    opt = D.FlatSGD(flat, lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True)
    with D.kernels.use_ops(torch_ops):
        for _ in range(2):
            opt.zero_grad()
            if gather:      # the bench's step: no collective inside train_step, gradients packed after backward
                out = model.train_step(batch_for(rank), None, sync_log_vars=False)
                assert all(torch.is_tensor(v) for v in out['log_vars'].values())
                out['loss'].backward()
                flat.collect_grads()
                log = D.reduce_log_vars(out['log_vars'])
            else:
                out = model.train_step(batch_for(rank), None)
                out['loss'].backward()
                log = out['log_vars']
            dp.allreduce_grads()
            opt.step()
    assert flat.check_views()
    # inference results come back in dataset order on every rank (rank r holds samples r, r+world, ...)
    part = [f'sample{i}' for i in range(rank, 5 + (5 % world and world - 5 % world), world)]
    assert D.gather_results(part, 5) == [f'sample{i}' for i in range(5)]
    torch.save(dict(p=flat.flat_p.clone(), g=flat.flat_g.clone(), log=log), os.path.join(out_dir, f'r{rank}.pt'))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize('gather', [False, True])
def test_two_rank_dp_equals_microbatch_average(tmp_path, gather):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), gather), nprocs=world, join=True)
    r0, r1 = [torch.load(tmp_path / f'r{r}.pt') for r in range(world)]
    assert torch.equal(r0['p'], r1['p'])                    # replicas stay bit-identical
    assert torch.equal(r0['g'], r1['g'])
    assert abs(r0['log']['loss'] - r1['log']['loss']) < 1e-12      # logged scalars are the all-reduced means

    # single-process restatement: rank-0 weights, two independent micro-batches, averaged gradients
    torch.set_num_threads(1)
    model = make_model(seed=7)
    flat = D.FlatParams(model)
    opt = D.FlatSGD(flat, lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True)
    states = [{k: v.clone() for k, v in model.state_dict().items() if 'running' in k or 'num_batches' in k}
              for _ in range(world)]
    with D.kernels.use_ops(torch_ops):
        for _ in range(2):
            opt.zero_grad()
            for r in range(world):                           # BN running stats are rank-local state
                model.load_state_dict(states[r], strict=False)
                (model.train_step(batch_for(r), None)['loss'] / world).backward()
                states[r] = {k: v.clone() for k, v in model.state_dict().items() if k in states[r]}
            opt.step()
    assert torch.allclose(flat.flat_p, r0['p'], rtol=0, atol=2e-6), (flat.flat_p - r0['p']).abs().max()


def _engine_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    from dsgcn_amd.engine import TrainEngine
    model = make_model(seed=7 + rank)
    calls = []
    real = {name: getattr(dist, name) for name in ('all_reduce', 'broadcast', 'all_gather', 'reduce', 'barrier',
                                                   'all_gather_into_tensor', 'reduce_scatter_tensor')}
    for name, fn in real.items():
        setattr(dist, name, (lambda fn, name: lambda *a, **k: (calls.append((name, a[0].numel() if a and torch.is_tensor(a[0]) else 0)),
                                                                fn(*a, **k))[1])(fn, name))
    try:
        with D.kernels.use_ops(torch_ops):
            eng = TrainEngine(model, lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True, use_graph=False)
            wrap = list(calls)
            del calls[:]
            b = batch_for(rank)
            for _ in range(3):
                eng.step(b['keypoint'], b['label'])
    finally:
        for name, fn in real.items():
            setattr(dist, name, fn)
    torch.save(dict(wrap=wrap, steps=list(calls), n=eng.flat.numel, p=eng.flat.flat_p.clone()), os.path.join(out_dir, f'e{rank}.pt'))
    dist.destroy_process_group()


def test_engine_step_is_one_collective():
    """TrainEngine itself (the object bench.py and train_model drive) under a 2-rank group, eager: wrapping
    costs the parameter broadcast + two packed buffer broadcasts, and a step is EXACTLY one collective — the all-reduce of
    the whole flat gradient buffer (no per-tensor buckets, no log-scalar reductions inside the step)."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_engine_worker, args=(2, _free_port(), tmp), nprocs=2, join=True)
        r0, r1 = [torch.load(os.path.join(tmp, f'e{r}.pt')) for r in range(2)]
    for r in (r0, r1):
        assert [c[0] for c in r['wrap']] == ['broadcast'] * 3 and r['wrap'][0][1] == r['n']
        assert r['steps'] == [('all_reduce', r['n'])] * 3
    assert torch.equal(r0['p'], r1['p'])


@pytest.mark.parametrize('gather', [False, True])
def test_flat_sgd_matches_torch_sgd(gather):
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 3), torch.nn.Linear(2, 2))   # last: no grad
    ref = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 3), torch.nn.Linear(2, 2))
    ref.load_state_dict(lin.state_dict())
    lin, ref = lin[:2], ref[:2]
    flat = D.FlatParams(lin, gather=gather)
    opt = D.FlatSGD(flat, lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True)
    topt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True)
    x = torch.randn(6, 5)
    for _ in range(3):
        opt.zero_grad()
        lin(x).square().sum().backward()
        flat.collect_grads()
        assert flat.check_views()
        opt.step()
        topt.zero_grad()
        ref(x).square().sum().backward()
        topt.step()
    for a, b in zip(lin.parameters(), ref.parameters()):
        assert torch.allclose(a, b, atol=1e-6)
    assert D.shard_batch(512, 3, 8) == (192, 256)
    assert abs(D.cosine_lr(0.1, 50, 100) - 0.05) < 1e-12


def test_bench_self_launch_spawns_before_any_gpu_call(monkeypatch):
    """``python bench.py --gpus N`` (N > 1, no launcher around it — the form the driver uses at N = 1) must become the
    PARENT of ``python -m torch.distributed.run`` before anything touches the GPU runtime: no
    torch.cuda.* call — not even is_available() — and a child process, never an exec.  The spawn is intercepted here; every
    torch.cuda entry point that initialises or queries the runtime is booby-trapped."""
    import subprocess
    import sys
    import bench

    def trap(name):
        def f(*a, **k):
            raise AssertionError(f'torch.cuda.{name} called before the launcher was spawned')
        return f
    for name in ('is_available', 'init', '_lazy_init', 'set_device', 'current_device', 'synchronize', 'device_count'):
        monkeypatch.setattr(torch.cuda, name, trap(name))
    for name in ('execv', 'execve', 'execvp', 'execvpe', 'execl', 'execlp'):
        monkeypatch.setattr(os, name, trap('os.' + name))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'DSGCN_BENCH_SELF_LAUNCH'):
        monkeypatch.delenv(k, raising=False)
    seen = {}

    class FakeChild:
        def __init__(self, cmd, env=None, cwd=None, stdout=None, text=None):
            seen.update(cmd=cmd, env=env, cwd=cwd)
            self.stdout = iter(['{"metric": "x", "n_gpus": 8}\n'])

        def wait(self):
            return 7

        def kill(self):
            seen['killed'] = True

    monkeypatch.setattr(subprocess, 'Popen', FakeChild)
    with pytest.raises(SystemExit) as exc:
        bench.main(['--gpus', '8', '--steps', '5', '--warmup', '2'])
    assert exc.value.code == 7                                   # the children's exit code is the parent's
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '8' and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ['--gpus', '8', '--steps', '5', '--warmup', '2']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and 'WORLD_SIZE' not in seen['env']
    assert seen['env']['DSGCN_BENCH_SELF_LAUNCHED'] == '1' and 'DSGCN_BENCH_FORCE_DIST' not in seen['env']
    # N = 1 through the same path (the GPU test's knob): a 1-rank RCCL group is forced in the child
    monkeypatch.setenv('DSGCN_BENCH_SELF_LAUNCH', '1')
    with pytest.raises(SystemExit):
        bench.main(['--steps', '1'])
    assert seen['env']['DSGCN_BENCH_FORCE_DIST'] == '1' and seen['cmd'][seen['cmd'].index('--nproc-per-node') + 1] == '1'
    # under a launcher (WORLD_SIZE set) nothing is spawned: the rank path runs — and stops at the first GPU call here
    monkeypatch.setenv('WORLD_SIZE', '2')
    monkeypatch.setenv('RANK', '0')
    seen.clear()
    with pytest.raises(AssertionError, match='is_available'):
        bench.main(['--gpus', '2'])
    assert not seen
