"""Round-4 golden fixture from the IMPORTED reference (build container only: needs /root/reference).

    python tests/golden/gen_golden_r4.py

  trajectory_stgcn_steplr.npz
      BASELINE config 1's schedule on a reduced-width vanilla ST-GCN (configs/stgcn/stgcn_vanilla_ntu60_xsub_3dkp/j.py:
      ``lr_config = dict(policy='step', step=[...])``, ``evaluation = dict(interval=1, metrics=['top_k_accuracy'])`` +
      ``mean_class_accuracy`` as the DS-GCN configs ask): four epochs of the reference model under torch SGD (momentum 0.9,
      nesterov, weight decay on every tensor), the rate set before each epoch the way mmcv's StepLrUpdaterHook(by_epoch=
      True) does (``base * gamma ** #{milestones <= epoch}``), the reference's own DistributedSampler order, and after every
      epoch the val split scored by the reference's ``forward_test`` (two clips per sample, ``average_clips='prob'``) and
      judged by the reference's ``top_k_accuracy`` / ``mean_class_accuracy`` (pyskl/core/evaluation.py:85-126).  Stored:
      initial state_dict, train / val data, sampler orders, per-iteration rate and loss, per-epoch val scores + metrics,
      final parameters (fp64 run = truth, the reference's own fp32 run = yardstick).
Data only (inputs and reference outputs); no reference source."""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from closed_form import liven32  # noqa: E402
from gen_golden import extract_feat_f64, other_cfg  # noqa: E402

R = ref_shim.load()
CFG = dict(samples=24, val_samples=12, val_clips=2, batch=8, epochs=4, seed=9, frames=16, lr=0.03, momentum=0.9,
           weight_decay=5e-4, classes=6, step=[1, 3], gamma=0.1, signal=2.0)


def step_rate(base, epoch, step, gamma):
    """mmcv StepLrUpdaterHook.get_lr with a milestone list, progress = runner.epoch (0-based, by_epoch=True)."""
    exp = len(step)
    for i, s in enumerate(step):
        if epoch < s:
            exp = i
            break
    return base * gamma ** exp


def model_cfg():
    cfg = other_cfg('stgcn', num_classes=CFG['classes'], base_channels=16, num_stages=4, inflate_stages=[3], down_stages=[3])
    cfg['cls_head']['in_channels'] = 32
    return cfg


def scores64(m64, x):
    """forward_test (recognizers/recognizergcn.py:41-107, average_clips='prob') on the fp64 copy."""
    N, clips = x.shape[:2]
    feats = extract_feat_f64(m64, x.double().flatten(0, 1))
    s = m64.cls_head(feats).view(N, clips, -1)
    return torch.softmax(s, 2).mean(1)


def main():
    cfg = model_cfg()
    T, V, classes = CFG['frames'], 25, CFG['classes']
    np.random.seed(4)
    torch.manual_seed(4)
    m32 = R.builder.build_model(cfg)
    liven32(m32, 41, 0.5)
    m32.test_cfg = dict(average_clips='prob')
    m64 = R.builder.build_model(cfg).double()
    m64.load_state_dict({k: v.double() if v.dtype.is_floating_point else v for k, v in m32.state_dict().items()})
    p0 = {k: p.detach().double().clone() for k, p in m32.named_parameters()}
    out = dict(config=np.array(json.dumps(CFG)), cfg=np.array(json.dumps(cfg)))
    for k, v in m32.state_dict().items():
        out['sd_' + k] = v.detach().numpy().copy()
    g = torch.Generator().manual_seed(78)
    # a class-dependent pattern over joints so that a few steps of training move the val metrics
    pat = torch.randn(classes, 1, 1, 1, V, 3, generator=g)
    ys = torch.randint(0, classes, (CFG['samples'],), generator=g)
    xs = torch.randn(CFG['samples'], 1, 2, T, V, 3, generator=g) + CFG['signal'] * pat[ys]
    yv = torch.arange(CFG['val_samples']) % classes
    xv = torch.randn(CFG['val_samples'], CFG['val_clips'], 2, T, V, 3, generator=g) + CFG['signal'] * pat[yv]
    out['x'], out['label'], out['xv'], out['label_v'] = xs.numpy(), ys.numpy(), xv.numpy(), yv.numpy()
    spec = importlib.util.spec_from_file_location('ref_sampler', os.path.join(ref_shim.REF_ROOT, 'pyskl', 'datasets',
                                                                               'samplers', 'distributed_sampler.py'))
    ref_sampler = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_sampler)
    sampler = ref_sampler.DistributedSampler(list(range(CFG['samples'])), num_replicas=1, rank=0, shuffle=True, seed=CFG['seed'])
    orders = []
    for ep in range(CFG['epochs']):
        sampler.set_epoch(ep)
        orders.append(list(iter(sampler)))
    out['order'] = np.array(orders)
    per_epoch = CFG['samples'] // CFG['batch']
    ev = R.evaluation
    for tag, m in (('32', m32), ('64', m64)):
        m.train()
        opt = torch.optim.SGD(m.parameters(), lr=CFG['lr'], momentum=CFG['momentum'], weight_decay=CFG['weight_decay'], nesterov=True)
        losses, lrs, vals, vscores = [], [], [], []
        for ep in range(CFG['epochs']):
            rate = step_rate(CFG['lr'], ep, CFG['step'], CFG['gamma'])       # before_train_epoch
            for grp in opt.param_groups:
                grp['lr'] = rate
            for b in range(per_epoch):
                idx = orders[ep][b * CFG['batch']:(b + 1) * CFG['batch']]
                x, y = xs[idx], ys[idx].view(-1, 1)
                opt.zero_grad()
                if tag == '32':
                    loss = m.train_step(dict(keypoint=x, label=y), opt)['loss']
                else:
                    logits = m.cls_head(extract_feat_f64(m, x[:, 0].double()))
                    loss = torch.nn.functional.cross_entropy(logits, y.squeeze(-1))
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
                lrs.append(rate)
            # after_train_epoch: the EvalHook (eval mode, running statistics; then back to train)
            m.eval()
            with torch.no_grad():
                if tag == '32':
                    sc = np.asarray(m(keypoint=xv, return_loss=False))
                else:
                    sc = scores64(m, xv).numpy()
            m.train()
            top = ev.top_k_accuracy(sc, yv.numpy(), (1, 5))
            vals.append([float(top[0]), float(top[1]), float(ev.mean_class_accuracy(sc, list(yv.numpy()))[0])])
            vscores.append(sc)
        out[f'loss{tag}'], out[f'lr{tag}'] = np.array(losses), np.array(lrs)
        out[f'val{tag}'] = np.array(vals)                      # (epochs, [top1, top5, mean_class_accuracy])
        out[f'vscores{tag}'] = np.stack(vscores).astype(np.float64)
    P32, P64 = dict(m32.named_parameters()), dict(m64.named_parameters())
    names = [k for k, p in P64.items() if p.grad is not None]
    out['names'] = np.array(json.dumps(names))
    for i, k in enumerate(names):
        out[f'p64_{i}'] = P64[k].detach().numpy()
    num = sum(float((P32[k].detach().double() - P64[k].detach()).pow(2).sum()) for k in names)
    den = sum(float(P64[k].detach().pow(2).sum()) for k in names)
    upd = sum(float((P64[k].detach() - p0[k]).pow(2).sum()) for k in names)
    out['perr32'], out['uerr32'] = np.array((num / den) ** .5), np.array((num / upd) ** .5)
    print('ref fp32 vs fp64: params', float(out['perr32']), 'update', float(out['uerr32']))
    print('lr', out['lr64'])
    print('val64', out['val64'])
    print('val32', out['val32'])
    print('score gap 32/64', np.abs(out['vscores32'] - out['vscores64']).max())
    np.savez_compressed(os.path.join(HERE, 'trajectory_stgcn_steplr.npz'), **out)
    print('wrote trajectory_stgcn_steplr.npz', os.path.getsize(os.path.join(HERE, 'trajectory_stgcn_steplr.npz')), 'bytes')


if __name__ == '__main__':
    main()
