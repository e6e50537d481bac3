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

``validate=True`` registers the counterpart of the reference's ``DistEvalHook`` (train.py:134-147 over mmcv's EvalHook):
every ``evaluation.interval`` epochs the val split goes through ``forward_test`` rank-sharded in dataset order, the parts
are gathered (``gather_results``), rank 0 computes ``evaluation.metrics`` (top-k / mean-class accuracy) and keeps
``best_<key>_epoch_<n>.pth`` (``save_best='auto'``: the first metric returned).  ``lr_config`` policies: ``CosineAnnealing``
by iteration (the DS-GCN configs), ``step`` by epoch (configs/stgcn/stgcn_vanilla_ntu60_xsub_3dkp/j.py:35), ``fixed``.

Not reproduced: TensorBoard logging, multi-optimizer configs, gradient clipping (the shipped configs set
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
from .evaluation import mean_class_accuracy, top_k_accuracy
from .recognizers import gather_results, reduce_log_vars
from .train import cosine_lr, step_lr


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
    """index list -> (keypoint (B, clips, M, T, V, C) float32, label (B, 1) int64) on the device.

    With the resident (``SkeletonStore``, ``SkeletonBatcher``) pair the host half of a batch (``plan``: RNG draws + index
    gathers) is made ONE BATCH AHEAD on a feeder thread while the device runs the current step: ``batch(idx, nxt)`` hands
    back the batch for ``idx`` and starts the plan for ``nxt``.  One worker, one plan at a time, in loop order: numpy's
    global RNG is consumed in exactly the order of the unthreaded loop."""

    def __init__(self, dataset, device, prefetch=True):
        self.device = device
        self.store = self.batcher = None
        self._pool = self._pending = self._pending_key = None
        if isinstance(dataset, (tuple, list)) and len(dataset) == 2 and hasattr(dataset[1], 'plan'):
            self.store, self.batcher = dataset
            self.n = len(self.store)
            if hasattr(self.batcher, 'prepare'):
                self.batcher.prepare(self.store)          # the per-clip decisions, once (first epoch at full rate)
            if prefetch:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix='dsgcn-feeder')
        else:
            self.dataset = dataset
            self.n = len(dataset)

    def __len__(self):
        return self.n

    def _plan(self, indices):
        key = tuple(indices)
        if self._pending is not None:
            fut, pkey = self._pending, self._pending_key
            self._pending = self._pending_key = None
            plan = fut.result()                           # (also orders the RNG: the feeder has finished its draws)
            if pkey == key:
                return plan
            raise RuntimeError('_BatchSource: the batch asked for is not the one announced as next')
        return self.batcher.plan(self.store, indices)

    def batch(self, indices, next_indices=None):
        if self.store is not None:
            plan = self._plan(indices)
            if self._pool is not None and next_indices is not None and len(next_indices):
                nxt = list(next_indices)
                self._pending_key = tuple(nxt)
                self._pending = self._pool.submit(self.batcher.plan, self.store, nxt)
            return self.batcher.run(self.store, plan)
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
        if lr_cfg.get('policy') not in ('CosineAnnealing', 'step', 'Step', 'fixed', 'Fixed'):
            raise NotImplementedError(f"lr_config policy {lr_cfg.get('policy')!r}: the skeleton configs use "
                                      'CosineAnnealing or step')
        if lr_cfg.get('by_epoch', True) and lr_cfg['policy'] == 'CosineAnnealing':
            raise NotImplementedError('CosineAnnealing by_epoch=True is not used by the skeleton configs')
        if lr_cfg['policy'] in ('step', 'Step') and not lr_cfg.get('by_epoch', True):
            raise NotImplementedError('step policy by_epoch=False is not used by the skeleton configs')
        if lr_cfg.get('warmup'):
            raise NotImplementedError('lr warm-up is not used by the skeleton configs')
        self.lr_cfg = lr_cfg
        self.log_interval = int((_get(cfg, 'log_config', None) or {}).get('interval', 20))
        ck = _get(cfg, 'checkpoint_config', None)          # None = no CheckpointHook (mmcv register_checkpoint_hook)
        self.ckpt_interval = 0 if ck is None else int(dict(ck).get('interval', 1))
        self.log = []
        self.evaluator = None                              # EvalLoop, set by train_model(validate=True)
        per_rank = int(math.ceil(len(source) / self.world))
        self.iters_per_epoch = per_rank // self.batch_size if self.drop_last else int(math.ceil(per_rank / self.batch_size))
        self.max_iters = self.max_epochs * self.iters_per_epoch

    def current_lr(self):
        base = self.engine.opt.base_lr
        if self.lr_cfg['policy'] == 'CosineAnnealing':
            return cosine_lr(base, self.iter, self.max_iters, float(self.lr_cfg.get('min_lr', 0) or 0))
        if self.lr_cfg['policy'] in ('step', 'Step'):     # mmcv StepLrUpdaterHook(by_epoch=True): set before each epoch
            return step_lr(base, self.epoch, self.lr_cfg['step'], float(self.lr_cfg.get('gamma', 0.1)),
                           self.lr_cfg.get('min_lr', None))
        return base

    def train_epoch(self):
        self.model.train()
        order = epoch_indices(len(self.source), self.epoch, self.seed, self.rank, self.world, self.shuffle)
        pending, t0 = [], time.perf_counter()
        for b in range(self.iters_per_epoch):
            idx = order[b * self.batch_size:(b + 1) * self.batch_size]
            nxt = order[(b + 1) * self.batch_size:(b + 2) * self.batch_size] if b + 1 < self.iters_per_epoch else None
            kp, lb = self.source.batch(idx, nxt)
            lr = self.current_lr()                                    # before_train_iter
            logs = self.engine.step(kp, lb, lr)                       # run_iter + after_train_iter (OptimizerHook)
            # a replayed hipGraph hands back the SAME static tensors every iteration: keep this iteration's values
            pending.append(({k: v.clone() for k, v in logs.items()}, len(idx)))
            self.iter += 1
            if (b + 1) % self.log_interval == 0 or b + 1 == self.iters_per_epoch:
                self._flush_log(pending, lr, b + 1, time.perf_counter() - t0)
                pending, t0 = [], time.perf_counter()
        self.epoch += 1
        if self.work_dir and self.ckpt_interval and self.epoch % self.ckpt_interval == 0 and self.rank == 0:
            self.save_checkpoint()
        if self.evaluator is not None:                                # after_train_epoch of the EvalHook
            self.evaluator.after_train_epoch(self)

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
        if meta.get('hook_msgs'):
            self.meta['hook_msgs'] = dict(meta['hook_msgs'])
        if self.evaluator is not None:
            self.evaluator.restore(self.meta)
        return meta

    def run(self):
        while self.epoch < self.max_epochs:
            self.train_epoch()
        return self


class EvalLoop:
    """The reference's ``DistEvalHook`` (pyskl/core/evaluation.py:11-19 over mmcv ``EvalHook`` / ``DistEvalHook``) for the
    epoch-based runner: after every ``interval``-th epoch (from ``start`` on) the val split is scored with ``forward_test``
    — rank r takes samples r, r + world, ... in dataset order (``DistributedSampler(shuffle=False)``), ``videos_per_gpu``
    of the val dataloader per call — the parts are interleaved back (``gather_results``), and rank 0 evaluates
    ``metrics`` the way ``BaseDataset.evaluate`` does (datasets/base.py:111-199: ``top_k_accuracy`` -> ``top{k}_acc``,
    ``mean_class_accuracy``).  ``save_best='auto'`` takes the first key returned; the rule (greater / less) follows the
    key name as in mmcv (``acc`` / ``top`` greater, ``loss`` less); the best checkpoint is ``best_<key>_epoch_<n>.pth`` and
    the previous best file is removed.  ``results`` holds one record per evaluation."""
    greater_keys = ('acc', 'top', 'AR@', 'auc', 'precision', 'mAP@', 'Recall@')
    less_keys = ('loss',)

    def __init__(self, dataset, batch_size=1, interval=1, start=None, metrics='top_k_accuracy',
                 metric_options=None, save_best='auto', rule=None, by_epoch=True, device='cuda',
                 broadcast_bn_buffer=True, **unsupported):
        if not by_epoch:
            raise NotImplementedError('evaluation by iteration is not used by the skeleton configs')
        unsupported.pop('key_indicator', None)
        unsupported.pop('gpu_collect', None)
        unsupported.pop('tmpdir', None)
        if unsupported:
            raise NotImplementedError(f'evaluation options {sorted(unsupported)} are not supported')
        self.source = _BatchSource(dataset, device)
        self.batch_size, self.interval, self.start = int(batch_size), int(interval), start
        self.metrics = list(metrics) if isinstance(metrics, (list, tuple)) else [metrics]
        for m in self.metrics:
            if m not in ('top_k_accuracy', 'mean_class_accuracy', 'confusion_matrix'):
                raise KeyError(f'metric {m} is not supported')
        self.metric_options = dict(metric_options or dict(top_k_accuracy=dict(topk=(1, 5))))
        self.save_best, self.rule = save_best, rule
        self.key_indicator = None if save_best in ('auto', None, True) else save_best
        self.best_score = self.best_ckpt = None
        self.broadcast_bn_buffer = bool(broadcast_bn_buffer)
        self.results = []

    def restore(self, meta):
        """After a resume: the best score / file so far ride in the checkpoint's ``meta['hook_msgs']`` (mmcv EvalHook reads
        them back in ``before_run`` / ``_save_ckpt``), so the first evaluation of the resumed run competes with them."""
        msgs = (meta or {}).get('hook_msgs') or {}
        if msgs.get('best_score') is not None:
            self.best_score, self.best_ckpt = msgs['best_score'], msgs.get('best_ckpt')
            self.key_indicator = msgs.get('key_indicator', self.key_indicator)

    @torch.no_grad()
    def sync_bn_buffers(self, model, world):
        """mmcv ``DistEvalHook._do_evaluate`` with ``broadcast_bn_buffer=True`` (its default): BatchNorm running statistics
        evolve rank-locally during training (no per-step buffer sync, pyskl/apis/train.py:98-102), so before a validation
        pass every rank takes rank 0's — the scores then describe the weights AND buffers rank 0 saves.  One packed
        collective for all layers."""
        if world <= 1 or not self.broadcast_bn_buffer:
            return
        bufs = []
        for m in model.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.track_running_stats:
                bufs += [m.running_var, m.running_mean]
        if not bufs:
            return
        packed = torch.cat([b.detach().reshape(-1).float() for b in bufs])
        dist.broadcast(packed, src=0)
        off = 0
        for b in bufs:
            b.copy_(packed[off:off + b.numel()].view_as(b).to(b.dtype))
            off += b.numel()

    def labels(self):
        if self.source.store is not None:
            return [int(v) for v in self.source.store.labels]
        ds = self.source.dataset
        if hasattr(ds, 'video_infos'):
            return [int(np.asarray(a['label']).reshape(-1)[0]) for a in ds.video_infos]
        return [int(np.asarray(ds[i]['label']).reshape(-1)[0]) for i in range(len(ds))]

    @torch.no_grad()
    def predict(self, model, rank, world):
        """This rank's share of the val scores, in its sampler order."""
        n = len(self.source)
        order = epoch_indices(n, 0, 0, rank, world, shuffle=False)
        was_training = model.training
        model.eval()
        part = []
        # the reference samples val clips in loader worker processes: the test-mode sampler's np.random.seed(255) never
        # touches the TRAINING process's stream.  Here both run in one process, so the stream is put back afterwards.
        rng = np.random.get_state()
        try:
            for b in range(0, len(order), self.batch_size):
                kp, _ = self.source.batch(order[b:b + self.batch_size])
                part.extend(model(keypoint=kp, return_loss=False))
        finally:
            np.random.set_state(rng)
        model.train(was_training)
        return part

    def evaluate(self, scores, labels):
        out = OrderedDict()
        for metric in self.metrics:
            if metric == 'top_k_accuracy':
                topk = self.metric_options.get('top_k_accuracy', {}).get('topk', (1, 5))
                topk = (topk,) if isinstance(topk, int) else tuple(topk)
                for k, acc in zip(topk, top_k_accuracy(scores, labels, topk)):
                    out[f'top{k}_acc'] = float(acc)
            elif metric == 'mean_class_accuracy':
                out['mean_class_accuracy'] = float(mean_class_accuracy(scores, labels)[0])
            elif metric == 'confusion_matrix':
                out['confusion_matrix'] = mean_class_accuracy(scores, labels)[1]
        return out

    def _better(self, key, value):
        if self.best_score is None:
            return True
        rule = self.rule
        if rule is None:
            low = key.lower()
            if any(k.lower() in low for k in self.greater_keys):
                rule = 'greater'
            elif any(k.lower() in low for k in self.less_keys):
                rule = 'less'
            else:
                raise ValueError(f'Cannot infer the rule for key {key}, thus a specific rule must be specified.')
        return value > self.best_score if rule == 'greater' else value < self.best_score

    def should_run(self, epoch):
        """``epoch`` = completed epochs (mmcv ``_should_evaluate``: every ``interval`` epochs, from ``start`` on)."""
        if self.start is None:
            return epoch % self.interval == 0
        return epoch >= self.start and (epoch - self.start) % self.interval == 0

    def after_train_epoch(self, runner):
        if not self.should_run(runner.epoch):
            return None
        self.sync_bn_buffers(runner.model, runner.world)
        part = self.predict(runner.model, runner.rank, runner.world)
        scores = gather_results(part, len(self.source))
        rec = None
        if runner.rank == 0:
            vals = self.evaluate(np.stack(scores), self.labels())
            rec = dict(epoch=runner.epoch, **vals)
            self.results.append(rec)
            if runner.logger is not None:
                runner.logger.info('Epoch(val) [%d]\t%s', runner.epoch,
                                   ', '.join(f'{k}: {v:.4f}' for k, v in vals.items() if np.ndim(v) == 0))
            if self.save_best and vals and runner.work_dir:
                key = self.key_indicator or next(iter(vals))
                self.key_indicator = key
                if self._better(key, vals[key]):
                    self.best_score = vals[key]
                    if self.best_ckpt and os.path.isfile(self.best_ckpt):
                        os.remove(self.best_ckpt)
                    os.makedirs(runner.work_dir, exist_ok=True)
                    self.best_ckpt = os.path.join(runner.work_dir, f'best_{key}_epoch_{runner.epoch}.pth')
                    # kept in the runner's meta as mmcv does: every later epoch_N.pth carries it, a resume reads it back
                    runner.meta['hook_msgs'] = dict(best_score=self.best_score, best_ckpt=self.best_ckpt, key_indicator=key)
                    meta = dict(runner.meta, epoch=runner.epoch, iter=runner.iter)
                    save_checkpoint(runner.model, self.best_ckpt, optimizer=runner.engine.opt, meta=meta)
        if runner.world > 1:
            dist.barrier()
        return rec


def train_model(model, dataset, cfg, distributed=None, validate=False, test=None, timestamp=None, meta=None,
                device='cuda', logger=None, use_graph=True, val_dataset=None, prefetch=True):
    """Train ``model`` on ``dataset`` the way the reference's ``train_model`` does for the skeleton configs; returns the
    ``EpochRunner`` (its ``.log`` holds the interval records, ``.engine`` the optimizer state).

    cfg (``Config`` or dict) keys read: ``data.videos_per_gpu`` / ``data.train_dataloader``, ``optimizer`` (SGD), ``lr_config``,
    ``total_epochs``, ``checkpoint_config``, ``log_config.interval``, ``work_dir``, ``seed``, ``resume_from`` / ``load_from`` /
    ``auto_resume``; with ``validate=True`` also ``evaluation`` and ``data.val`` / ``data.val_dataloader`` (``val_dataset``
    overrides ``data.val``: a map-style dataset or a (``SkeletonStore``, ``SkeletonBatcher``) pair built for the val
    pipeline).  ``runner.evaluator.results`` holds the evaluation records.  ``prefetch``: plan the next batch of a resident
    store on a feeder thread while the device runs the current step (same RNG stream either way)."""
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
    source = _BatchSource(dataset, next(model.parameters()).device, prefetch=prefetch)
    work_dir = _get(cfg, 'work_dir', None)
    runner = EpochRunner(model, engine, source, cfg, work_dir=work_dir, meta=meta, logger=logger)
    if validate:
        data = _get(cfg, 'data', {}) or {}
        if val_dataset is None:
            from .pipeline import build_dataset
            val_cfg = dict(data['val'])
            val_cfg.setdefault('test_mode', True)
            val_dataset = build_dataset(val_cfg)
        vloader = dict(videos_per_gpu=data.get('videos_per_gpu', 1))
        vloader.update(data.get('val_dataloader', {}) or {})
        eval_cfg = dict(_get(cfg, 'evaluation', None) or {})
        runner.evaluator = EvalLoop(val_dataset, batch_size=vloader['videos_per_gpu'],
                                    device=next(model.parameters()).device, **eval_cfg)
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
