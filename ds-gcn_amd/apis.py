"""``train_model`` behind the reference's entry point (pyskl/apis/train.py:52-223) for the skeleton configs.

What the reference assembles out of mmcv parts — ``MMDistributedDataParallel`` (train.py:94-102), ``build_optimizers``,
``EpochBasedSparseRunner`` (core/local_runner/epoch_based_sparse_runner.py:26-105), ``OptimizerHook``,
``CosineAnnealingLrUpdaterHook(by_epoch=False)``, ``CheckpointHook(interval)``, ``DistSamplerSeedHook``, the text logger and
resume / load_from (train.py:153-157) — is one loop here over ``engine.TrainEngine`` (two hipGraph replays + one RCCL
all-reduce per iteration):

    for epoch:  order = DistributedSampler(seed + epoch)            # datasets/samplers/distributed_sampler.py:27-43
        for batch in order:  lr = cosine(iter / max_iters)           # before_train_iter
                             engine.step(batch, lr)                  # train_step + backward + all-reduce + SGD
                             log every `interval` (ONE packed all-reduce + ONE host read)
        checkpoint every `checkpoint_config.interval` epochs         # epoch_{n}.pth + latest.pth, mmcv layout

Batches come from the HIP input pipeline when the dataset is a (``SkeletonStore``, ``SkeletonBatcher``) pair — clips
resident in HBM, host decisions, one launch per batch — or from any map-style dataset whose items are
``dict(keypoint=(clips, M, T, V, C), label=int)`` (the output of the reference-style ``Compose`` pipelines).

Not reproduced: validation hooks during training (``validate=True`` raises: run ``RecognizerGCN.forward_test`` over the
val split after training), TensorBoard logging, multi-optimizer configs, gradient clipping (the shipped configs set
``grad_clip=None``, configs/_init_/lr_schedual.py:12).
"""
import math
import os
import time
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist

from .checkpoint import find_resume, load_checkpoint, resume, save_checkpoint
from .engine import TrainEngine
from .recognizers import reduce_log_vars
from .train import cosine_lr


def _rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def epoch_indices(n, epoch, seed, rank, world, shuffle=True):
    """The reference's ``DistributedSampler.__iter__``: a permutation seeded with ``epoch + seed``, padded by wrapping to a
    multiple of the world size, every ``world``-th element from ``rank``."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(epoch + (seed if seed is not None else 0))
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    total = int(math.ceil(n / world)) * world
    idx += idx[:total - len(idx)]
    return idx[rank:total:world]


class _BatchSource:
    """index list -> (keypoint (B, clips, M, T, V, C) float32, label (B, 1) int64) on the device."""

    def __init__(self, dataset, device):
        self.device = device
        self.store = self.batcher = None
        if isinstance(dataset, (tuple, list)) and len(dataset) == 2 and hasattr(dataset[1], 'plan'):
            self.store, self.batcher = dataset
            self.n = len(self.store)
        else:
            self.dataset = dataset
            self.n = len(dataset)

    def __len__(self):
        return self.n

    def batch(self, indices):
        if self.store is not None:
            return self.batcher(self.store, indices)
        items = [self.dataset[i] for i in indices]
        kp = torch.stack([torch.as_tensor(np.asarray(it['keypoint']), dtype=torch.float32) for it in items])
        lb = torch.as_tensor([int(np.asarray(it['label']).reshape(-1)[0]) for it in items], dtype=torch.int64).view(-1, 1)
        return kp.to(self.device, non_blocking=True), lb.to(self.device, non_blocking=True)


class EpochRunner:
    """State of a run (``epoch``, ``iter``, ``max_iters``) + the loop; ``log`` collects one dict per logging interval."""

    def __init__(self, model, engine, source, cfg, work_dir=None, meta=None, logger=None):
        self.model, self.engine, self.source, self.cfg = model, engine, source, cfg
        self.work_dir, self.meta, self.logger = work_dir, dict(meta or {}), logger
        self.epoch = self.iter = 0
        self.rank, self.world = _rank_world()
        data = _get(cfg, 'data', {}) or {}
        loader = dict(data.get('train_dataloader', {}) or {})
        self.batch_size = int(loader.get('videos_per_gpu', data.get('videos_per_gpu', 1)))
        self.shuffle = bool(loader.get('shuffle', True))
        self.drop_last = bool(loader.get('drop_last', False))
        self.seed = _get(cfg, 'seed', None)
        self.max_epochs = int(_get(cfg, 'total_epochs', 1))
        lr_cfg = dict(_get(cfg, 'lr_config', None) or dict(policy='CosineAnnealing', min_lr=0, by_epoch=False))
        if lr_cfg.get('policy') not in ('CosineAnnealing', 'fixed', 'Fixed'):
            raise NotImplementedError(f"lr_config policy {lr_cfg.get('policy')!r}: the skeleton configs use CosineAnnealing")
        if lr_cfg.get('by_epoch', True) and lr_cfg['policy'] == 'CosineAnnealing':
            raise NotImplementedError('CosineAnnealing by_epoch=True is not used by the skeleton configs')
        self.lr_cfg = lr_cfg
        self.log_interval = int((_get(cfg, 'log_config', None) or {}).get('interval', 20))
        ck = _get(cfg, 'checkpoint_config', None) or {}
        self.ckpt_interval = int(ck.get('interval', 1)) if ck is not None else 0
        self.log = []
        per_rank = int(math.ceil(len(source) / self.world))
        self.iters_per_epoch = per_rank // self.batch_size if self.drop_last else int(math.ceil(per_rank / self.batch_size))
        self.max_iters = self.max_epochs * self.iters_per_epoch

    def current_lr(self):
        base = self.engine.opt.base_lr
        if self.lr_cfg['policy'] == 'CosineAnnealing':
            return cosine_lr(base, self.iter, self.max_iters, float(self.lr_cfg.get('min_lr', 0) or 0))
        return base

    def train_epoch(self):
        self.model.train()
        order = epoch_indices(len(self.source), self.epoch, self.seed, self.rank, self.world, self.shuffle)
        pending, t0 = [], time.perf_counter()
        for b in range(self.iters_per_epoch):
            idx = order[b * self.batch_size:(b + 1) * self.batch_size]
            kp, lb = self.source.batch(idx)
            lr = self.current_lr()                                    # before_train_iter
            pending.append((self.engine.step(kp, lb, lr), len(idx)))  # run_iter + after_train_iter (OptimizerHook)
            self.iter += 1
            if (b + 1) % self.log_interval == 0 or b + 1 == self.iters_per_epoch:
                self._flush_log(pending, lr, b + 1, time.perf_counter() - t0)
                pending, t0 = [], time.perf_counter()
        self.epoch += 1
        if self.work_dir and self.ckpt_interval and self.epoch % self.ckpt_interval == 0 and self.rank == 0:
            self.save_checkpoint()

    def _flush_log(self, pending, lr, inner, dt):
        """The interval's log scalars: sample-weighted means over its iterations, averaged over ranks with ONE collective
        and read back with ONE copy (the reference pays four all-reduces and four ``.item()`` syncs per ITERATION,
        recognizers/base.py:150-156)."""
        if not pending:
            return
        keys = list(pending[0][0])
        w = torch.tensor([n for _, n in pending], dtype=torch.float64, device=pending[0][0][keys[0]].device)
        means = OrderedDict((k, (torch.stack([lv[k].double() for lv, _ in pending]) * w).sum() / w.sum()) for k in keys)
        vals = reduce_log_vars(means)
        rec = dict(epoch=self.epoch + 1, iter=inner, lr=lr, time=dt / max(len(pending), 1), **vals)
        self.log.append(rec)
        if self.logger is not None and self.rank == 0:
            self.logger.info('Epoch [%d][%d/%d]\tlr: %.3e, time: %.3f, %s', rec['epoch'], inner, self.iters_per_epoch, lr,
                             rec['time'], ', '.join(f'{k}: {v:.4f}' for k, v in vals.items()))

    def save_checkpoint(self):
        os.makedirs(self.work_dir, exist_ok=True)
        meta = dict(self.meta, epoch=self.epoch, iter=self.iter)
        path = os.path.join(self.work_dir, f'epoch_{self.epoch}.pth')
        return save_checkpoint(self.model, path, optimizer=self.engine.opt, meta=meta, create_symlink=True)

    def resume(self, path):
        meta = resume(self.model, self.engine.opt, path)
        self.epoch, self.iter = int(meta.get('epoch', 0)), int(meta.get('iter', 0))
        return meta

    def run(self):
        while self.epoch < self.max_epochs:
            self.train_epoch()
        return self


def train_model(model, dataset, cfg, distributed=None, validate=False, test=None, timestamp=None, meta=None,
                device='cuda', logger=None, use_graph=True):
    """Train ``model`` on ``dataset`` the way the reference's ``train_model`` does for the skeleton configs; returns the
    ``EpochRunner`` (its ``.log`` holds the interval records, ``.engine`` the optimizer state).

    cfg (``Config`` or dict) keys read: ``data.videos_per_gpu`` / ``data.train_dataloader``, ``optimizer`` (SGD), ``lr_config``,
    ``total_epochs``, ``checkpoint_config``, ``log_config.interval``, ``work_dir``, ``seed``, ``resume_from`` / ``load_from`` /
    ``auto_resume``."""
    if validate:
        raise NotImplementedError('evaluation hooks during training are outside this path: run forward_test after training')
    if isinstance(dataset, list) and len(dataset) == 1:
        dataset = dataset[0]
    opt_cfg = dict(_get(cfg, 'optimizer', None) or dict(type='SGD', lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True))
    if opt_cfg.pop('type', 'SGD') != 'SGD':
        raise NotImplementedError('the skeleton configs train with SGD (configs/_init_/lr_schedual.py:11)')
    oc = _get(cfg, 'optimizer_config', None) or {}
    if oc.get('grad_clip'):
        raise NotImplementedError('grad_clip is None in the shipped configs; clipping is not implemented')
    model = model.to(device)
    world = _rank_world()[1]
    engine = TrainEngine(model, lr=opt_cfg.get('lr', 0.1), momentum=opt_cfg.get('momentum', 0),
                         weight_decay=opt_cfg.get('weight_decay', 0), nesterov=opt_cfg.get('nesterov', False),
                         use_graph=use_graph, strict_graph=world > 1)
    source = _BatchSource(dataset, next(model.parameters()).device)
    work_dir = _get(cfg, 'work_dir', None)
    runner = EpochRunner(model, engine, source, cfg, work_dir=work_dir, meta=meta, logger=logger)
    ckpt = find_resume(work_dir, _get(cfg, 'resume_from', None), bool(_get(cfg, 'auto_resume', False))) if work_dir else \
        _get(cfg, 'resume_from', None)
    if ckpt:
        runner.resume(ckpt)
    elif _get(cfg, 'load_from', None):
        load_checkpoint(model, _get(cfg, 'load_from'), strict=False)
    runner.run()
    if world > 1:
        dist.barrier()
    return runner
