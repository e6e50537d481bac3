"""Round-3 golden fixtures from the IMPORTED reference (build container only: needs /root/reference).

    python tests/golden/gen_golden_r3.py [units] [stgcn] [trajectory]

Adds to the earlier generators' sets (which it leaves untouched):
  unit_ds_r3.npz            the two headline units — dgphgcn1 (gcn.py:2074-2372) and dgmstcn (tcn.py:344-431) — at EVERY
                            width DS-STGCN uses (3->64 ... 256->256, stride 1 / 2, NTU and coco graphs), rebuilt on both
                            sides from the same seeded recipe (closed_form.make_ds_unit; the weights' digest is stored):
                            fp64 output, input gradient, EVERY parameter gradient (tensors > 16384 elements: their norm)
                            and the BatchNorm running statistics after the call;
  model_reduced_stgcn_shipped.npz / _cfg.json
                            the reference's shipped ST-GCN config (configs/stgcn/STGCN_model.py: unit_gcn + unitmlp) at
                            reduced width, same content as model_reduced_*.npz;
  trajectory_dsstgcn_reduced.npz
                            FIVE optimisation steps of a reduced-width DS-STGCN the way the reference's runner drives
                            them (configs/_init_/lr_schedual.py:11-27: torch SGD momentum 0.9 nesterov wd 5e-4 on every
                            parameter, cosine LR per iteration; BN running statistics moving): initial state_dict, the
                            five batches, per-step loss, whole-model parameter norms, the final parameters and BN running
                            statistics — fp64 run = truth, the reference's own fp32 run = yardstick (see trajectory()).
Data only (inputs and reference outputs); no reference source."""
import json
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from closed_form import DS_UNIT_CASES, liven32, make_ds_unit, pick_tensors, sd_digest, step_input  # noqa: E402
from gen_golden import ds_cfg, extract_feat_f64, other_cfg, reduced_model  # noqa: E402

R = ref_shim.load()


def f32(t):
    return t.detach().numpy().astype(np.float32)


def unit_ds():
    out = {}
    for tag in DS_UNIT_CASES:
        m, x, Rm = make_ds_unit(R.gutils, R.graph.Graph, tag)
        out[f'{tag}_digest'] = np.array(sd_digest(m))
        m = m.double().train()
        x = x.double().requires_grad_()
        y = m(x)
        (y * Rm.double()).sum().backward()
        out[f'{tag}_y'] = f32(y)
        out[f'{tag}_dx'] = f32(x.grad)
        for k, p in m.named_parameters():
            if p.grad is None:
                continue
            if p.numel() <= 16384:
                out[f'{tag}_grad_{k}'] = f32(p.grad)
            else:
                out[f'{tag}_gnorm_{k}'] = np.array(float(p.grad.norm()))
        for k, v in m.state_dict().items():
            if 'running' in k:
                out[f'{tag}_{k}'] = f32(v)
        print(tag, 'y', tuple(y.shape), 'grads', sum(1 for k in out if k.startswith(f'{tag}_grad_')),
              'norms', sum(1 for k in out if k.startswith(f'{tag}_gnorm_')))
    np.savez_compressed(os.path.join(HERE, 'unit_ds_r3.npz'), **out)


def stgcn_shipped():
    cfg = other_cfg('stgcn_shipped', 12, base_channels=16, num_stages=4, inflate_stages=[3], down_stages=[3])
    cfg['cls_head']['in_channels'] = 32
    reduced_model(cfg, 'model_reduced_stgcn_shipped', seed=4)


TRAJ = dict(samples=24, batch=8, epochs=2, seed=5, frames=16, lr=0.01, momentum=0.9, weight_decay=5e-4, classes=12)


def cosine(base_lr, it, total, min_lr=0.0):
    """mmcv CosineAnnealingLrUpdaterHook(by_epoch=False): annealing_cos(base_lr, min_lr, iter / max_iters)."""
    return min_lr + 0.5 * (base_lr - min_lr) * (1 + math.cos(math.pi * it / total))


def trajectory_cfg():
    cfg = ds_cfg(num_classes=TRAJ['classes'], base_channels=16, num_stages=4, inflate_stages=[3], down_stages=[3])
    cfg['cls_head']['in_channels'] = 32
    return cfg


def trajectory():
    """Why the reduced width and lr 0.01: a multi-step trajectory of the full-width, randomly initialised network is not a
    function fp32 can pin — the reference's OWN fp32 and fp64 runs (8 clips, lr 0.01) already disagree by 7 % on the
    gradient of step 2 and 32 % on step 3 (a 2e-6 parameter difference is amplified ~3e4 times by the batch-statistics
    backward under the mean-pooled head), by 46 % on the 5-step update, and at the shipped lr 0.1 by 3.5 % on the loss.
    At this width and rate its two runs stay 3e-6 apart on the parameters and 1e-3 on the update, so every ingredient of
    the loop (nesterov, weight decay on every tensor, cosine rate per iteration, BN momentum) is visible above the noise."""
    cfg = trajectory_cfg()
    T, V, classes = TRAJ['frames'], 25, TRAJ['classes']
    np.random.seed(3)
    torch.manual_seed(3)
    m32 = R.builder.build_model(cfg)
    liven32(m32, 33, 0.5)
    m64 = R.builder.build_model(cfg).double()
    m64.load_state_dict({k: v.double() if v.dtype.is_floating_point else v for k, v in m32.state_dict().items()})
    p0 = {k: p.detach().double().clone() for k, p in m32.named_parameters()}
    out = dict(config=np.array(json.dumps(TRAJ)), cfg=np.array(json.dumps(cfg)))
    for k, v in m32.state_dict().items():
        out['sd_' + k] = v.detach().numpy().copy()
    g = torch.Generator().manual_seed(77)
    xs = torch.randn(TRAJ['samples'], 1, 2, T, V, 3, generator=g)            # the dataset: 24 one-clip samples
    ys = torch.randint(0, classes, (TRAJ['samples'],), generator=g)
    out['x'], out['label'] = xs.numpy(), ys.numpy()
    # the visiting order comes from the reference's own sampler (datasets/samplers/distributed_sampler.py, torch only)
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_sampler', os.path.join(ref_shim.REF_ROOT, 'pyskl', 'datasets',
                                                                               'samplers', 'distributed_sampler.py'))
    ref_sampler = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_sampler)
    sampler = ref_sampler.DistributedSampler(list(range(TRAJ['samples'])), num_replicas=1, rank=0, shuffle=True,
                                             seed=TRAJ['seed'])
    orders = []
    for ep in range(TRAJ['epochs']):
        sampler.set_epoch(ep)                                                 # DistSamplerSeedHook
        orders.append(list(iter(sampler)))
    out['order'] = np.array(orders)
    per_epoch = TRAJ['samples'] // TRAJ['batch']
    total_iters = TRAJ['epochs'] * per_epoch
    runs = {}
    for tag, m in (('32', m32), ('64', m64)):
        m.train()
        opt = torch.optim.SGD(m.parameters(), lr=TRAJ['lr'], momentum=TRAJ['momentum'], weight_decay=TRAJ['weight_decay'],
                              nesterov=True)
        losses, pnorms, it = [], [], 0
        for ep in range(TRAJ['epochs']):
            for b in range(per_epoch):
                idx = orders[ep][b * TRAJ['batch']:(b + 1) * TRAJ['batch']]
                for grp in opt.param_groups:                             # before_train_iter: cosine LR per iteration
                    grp['lr'] = cosine(TRAJ['lr'], it, total_iters)
                x, y = xs[idx], ys[idx].view(-1, 1)
                opt.zero_grad()
                if tag == '32':
                    loss = m.train_step(dict(keypoint=x, label=y), opt)['loss']   # what the runner calls (run_iter)
                else:
                    logits = m.cls_head(extract_feat_f64(m, x[:, 0].double()))
                    loss = torch.nn.functional.cross_entropy(logits, y.squeeze(-1))
                loss.backward()
                opt.step()
                it += 1
                losses.append(float(loss.detach()))
                pnorms.append(float(sum(p.detach().double().pow(2).sum() for p in m.parameters()) ** .5))
            if ep == 0 and tag == '64':                                  # state at the epoch boundary (checkpoint / resume test)
                out['pnorm64_epoch1'] = np.array(pnorms[-1])
        runs[tag] = losses
        out[f'loss{tag}'] = np.array(losses)
        out[f'pnorm{tag}'] = np.array(pnorms)
    P32, P64 = dict(m32.named_parameters()), dict(m64.named_parameters())
    names = [k for k, p in P64.items() if p.grad is not None]
    out['names'] = np.array(json.dumps(names))
    for i, k in enumerate(names):
        out[f'p64_{i}'] = P64[k].detach().numpy()                        # fp64: the update is ~3e-3 of the parameter
    num = sum(float((P32[k].detach().double() - P64[k].detach()).pow(2).sum()) for k in names)
    den = sum(float(P64[k].detach().pow(2).sum()) for k in names)
    upd = sum(float((P64[k].detach() - p0[k]).pow(2).sum()) for k in names)
    out['perr32'] = np.array((num / den) ** .5)
    out['uerr32'] = np.array((num / upd) ** .5)
    sd64 = m64.state_dict()
    rk = [k for k in sd64 if k.endswith(('running_mean', 'running_var'))]
    out['running_names'] = np.array(json.dumps(rk))
    for i, k in enumerate(rk):
        out[f'running64_{i}'] = f32(sd64[k])
    out['nbt'] = np.array(int(m32.state_dict()['backbone.data_bn.num_batches_tracked']))
    print('ref fp32 vs fp64 after 6 steps: params', float(out['perr32']), 'update', float(out['uerr32']),
          'loss', [abs(a - b) / abs(b) for a, b in zip(runs['32'], runs['64'])])
    np.savez_compressed(os.path.join(HERE, 'trajectory_dsstgcn_reduced.npz'), **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['units', 'stgcn', 'trajectory']
    if 'units' in which:
        unit_ds()
    if 'stgcn' in which:
        stgcn_shipped()
    if 'trajectory' in which:
        trajectory()
    for fn in sorted(os.listdir(HERE)):
        if fn.startswith(('unit_ds_r3', 'model_reduced_stgcn_shipped', 'trajectory_')):
            print(fn, os.path.getsize(os.path.join(HERE, fn)))
