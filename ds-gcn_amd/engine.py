"""One optimisation step of a recognizer on one GPU of a data-parallel job, the MI355X way.

The reference's step (mmcv runner over torch DDP: pyskl/core/local_runner/epoch_based_sparse_runner.py:26-52 with
OptimizerHook and CosineAnnealingLrUpdaterHook, pyskl/apis/train.py:94-134) is ~2 k eager launches, four host syncs for
the log scalars and a bucketed NCCL all-reduce.  Here:

    graph A (hipGraph replay): zero-grad + forward + backward + pack gradients into the flat buffer
    one RCCL all-reduce of the flat gradient buffer (N > 1; between the graphs, never captured)
    graph B (hipGraph replay): SGD-nesterov update on the flat buffers, rate read from a device scalar

``bench.py`` times exactly this object, ``apis.train_model`` drives it epoch by epoch.
"""
import torch
import torch.distributed as dist

from . import kernels
from .data_parallel import FlatDataParallel, FlatParams
from .train import FlatSGD


class TrainEngine:

    def __init__(self, model, lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True, process_group=None, use_graph=True,
                 warmup_eager=2, strict_graph=False, extra_allreduce=False):
        """strict_graph: a failed capture raises instead of falling back to eager launches (multi-GPU runs: ranks must not
        silently differ).  extra_allreduce: issue the gradient all-reduce even at world size 1 (measurement of the N > 1
        call sequence under a 1-rank RCCL group)."""
        self.model = model
        self.flat = FlatParams(model, gather=True)
        self.dp = FlatDataParallel(self.flat, process_group)
        self.opt = FlatSGD(self.flat, lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov,
                           capturable=True)
        self.use_graph = bool(use_graph) and self.flat.flat_p.is_cuda
        self.strict_graph = strict_graph
        self.extra_allreduce = extra_allreduce
        self.warmup_eager = warmup_eager
        self._graphs = {}          # batch shape -> (graph A, graph B, static keypoint, static label, static outputs)
        self._seen = {}            # batch shape -> eager steps taken
        self.capture_error = None
        self._seed = None
        self.iter = 0

    # ---- pieces --------------------------------------------------------------------------------------------
    def _fwd_bwd(self, keypoint, label):
        self.opt.zero_grad()
        kernels.reset_leaf_uses()
        try:
            out = self.model.train_step(dict(keypoint=keypoint, label=label), None, sync_log_vars=False)
            loss = out['loss']
            if self._seed is None or self._seed.dtype != loss.dtype or self._seed.device != loss.device:
                self._seed = torch.ones((), dtype=loss.dtype, device=loss.device)  # (backward() fills a fresh one per step)
            if self.flat.flat_p.is_cuda:
                # parameter-gradient partial rows are summed by ONE launch at the end of the backward (kernels.param_colsum)
                with kernels.deferred_param_sums(self.flat):
                    loss.backward(self._seed)
            else:
                loss.backward(self._seed)
            self.flat.collect_grads()
            kernels.dropout_step_advance()       # (fused dropout: the next step draws other masks; no-op without it)
        finally:
            # the update that follows rewrites the weights through raw pointers: cached weight images are stale from here
            kernels.end_step()
        return {k: v.detach() for k, v in out['log_vars'].items()}

    def _exchange(self):
        self.dp.allreduce_grads()
        if self.extra_allreduce and self.dp.world == 1 and dist.is_available() and dist.is_initialized():
            dist.all_reduce(self.flat.flat_g)

    def _capture(self, keypoint, label):
        # no extra warm-up pass here: the eager steps that precede the capture (warmup_eager) already ran this shape —
        # allocator pools, the pinned pointer table of dsgcn_pack, the momentum buffer — and a pass that is not a real
        # step would move the BatchNorm running statistics once too often
        skp, slb = keypoint.clone(), label.clone()
        torch.cuda.synchronize()
        g_a = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread polls its events while we capture (N > 1); neither graph holds a collective
        with torch.cuda.graph(g_a, capture_error_mode='thread_local'):
            logs = self._fwd_bwd(skp, slb)
        g_b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_b, capture_error_mode='thread_local'):
            self.opt.step()
        torch.cuda.synchronize()
        return g_a, g_b, skp, slb, logs

    # ---- the step ------------------------------------------------------------------------------------------
    def step(self, keypoint, label, lr=None):
        """-> dict of detached DEVICE scalars (loss, loss_cls, top1_acc, top5_acc): no host sync here."""
        if lr is not None:
            self.opt.set_lr(lr)
        key = (tuple(keypoint.shape), tuple(label.shape))
        if self.use_graph and key not in self._graphs and self._seen.get(key, 0) >= self.warmup_eager:
            try:
                self._graphs[key] = self._capture(keypoint, label)
            except Exception as exc:       # noqa: BLE001 — report and fall back (or raise) below
                self.capture_error = f'{type(exc).__name__}: {exc}'
                if self.strict_graph:
                    raise
                self.use_graph = False
        if self.use_graph and key in self._graphs:
            g_a, g_b, skp, slb, logs = self._graphs[key]
            skp.copy_(keypoint)
            slb.copy_(label)
            g_a.replay()
            self._exchange()
            g_b.replay()
        else:
            logs = self._fwd_bwd(keypoint, label)
            self._exchange()
            self.opt.step()
            self._seen[key] = self._seen.get(key, 0) + 1
        self.iter += 1
        return logs

    def graphed(self, keypoint, label):
        return (tuple(keypoint.shape), tuple(label.shape)) in self._graphs
