"""Fused ops of the DS-GCN hot path, each a ``torch.autograd.Function`` over the C-ABI HIP library.

PyTorch is plumbing here (device memory, the current stream, autograd bookkeeping); every byte of
arithmetic in these ops happens in ``libdsgcn.so`` (``ds-gcn_amd/csrc``).  There is NO CPU or
eager fallback: inputs must be CUDA fp32 tensors and the library must load, otherwise the op raises.

``ops()`` returns the active op namespace (this module).  ``use_ops(ns)`` is a test seam that swaps
in another namespace with the same signatures (``tests/torch_ops.py``) so the host-side wiring can be
checked against the oracle without a GPU; the product never calls it.
"""
import contextlib
import sys

_active = None


def ops():
    return _active if _active is not None else sys.modules[__name__]


@contextlib.contextmanager
def use_ops(namespace):
    global _active
    prev = _active
    _active = namespace
    try:
        yield namespace
    finally:
        _active = prev


NAME = 'hip'

import torch

from . import native


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('dsgcn: the hot path runs only as HIP kernels on CUDA/ROCm tensors; got a CPU tensor '
                               '(there is no CPU fallback)')


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


_side_streams = {}
import os as _os
# K-B on a second stream beside the `pre` channel mix (side_branch below).  Off by default since round 3: with one
# workgroup per sample K-B left half of the chip to the main stream and the overlap paid (round 2); as (sample, subset)
# workgroups on the matrix core it fills the chip by itself and the fork / join edges only cost (13.33 ms/step with the
# side stream, 13.07 without; profiles/r03/README.md)
OVERLAP = _os.environ.get('DSGCN_OVERLAP', '0') != '0'
# csrc/tms.hip, the fused temporal stage: 'auto' = where it measured faster than the staged chain (5-tap windows: CTR-GCN's
# MSTCN, 22.3 -> 21.0 ms/step; on DS-STGCN's 3-tap units the staged wide-load kernels are still ahead, 13.6 vs 16.8 ms:
# profiles/r03/README.md), '1' = wherever eligible, '0' = never
FUSED_TEMPORAL = _os.environ.get('DSGCN_FUSED_TEMPORAL', 'auto')
# dgmstcn units (kernel 3) on the split layout of csrc/tmsplit.hip: '2' (default) = stride 1 and 2, '1' = stride 1 only,
# '0' = the staged three-kernel form everywhere
SPLIT_TEMPORAL = _os.environ.get('DSGCN_SPLIT_TEMPORAL', '2')
# '1': a block whose stride-2 residual conv reads the plain block input gets the even frames from the previous block's
# fuse_out (a second output of that launch) instead of a strided-copy launch of its own
PRESTRIDED = _os.environ.get('DSGCN_PRESTRIDED', '1') != '0'
# '1': the small ends of the step on csrc/head.hip (head + loss + accuracies in three launches, all BatchNorm buffers in
# one); '0': the framework's own launches (A/B switch)
FUSED_ENDS = _os.environ.get('DSGCN_FUSED_ENDS', '1') != '0'


class side_branch:
    """``with side_branch(t): ...`` runs the body on a second HIP stream that first waits for the current one, and
    joins it back on exit — the forked work overlaps whatever the main stream launches until ``join()``.  Autograd
    replays each node's backward on the stream of its forward, so the backward overlaps the same way.  Capturable in
    a hipGraph (fork/join become graph edges)."""

    def __init__(self, like):
        self.enabled = OVERLAP and like.is_cuda
        if self.enabled:
            dev = like.device
            if dev not in _side_streams:
                _side_streams[dev] = torch.cuda.Stream(device=dev)
            self.side = _side_streams[dev]
            self.main = torch.cuda.current_stream(dev)
            if self.main == self.side:            # already inside a forked region (e.g. K-B's backward): no nesting
                self.enabled = False

    def __enter__(self):
        if self.enabled:
            self.side.wait_stream(self.main)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.enabled:
            self.ctx.__exit__(*exc)
        return False

    def join(self, *tensors):
        """Make the main stream wait for the side work; ``tensors`` were produced on the side stream."""
        if self.enabled:
            self.main.wait_stream(self.side)
            for t in tensors:
                t.record_stream(self.main)


def _colsum_raw(t, R, C, inner=1):
    out = torch.empty(C, device=t.device, dtype=torch.float32)
    if inner > 1:
        native.check(native.lib().dsgcn_colsum_t(_ptr(t), R, C, inner, _ptr(out), _stream()), 'dsgcn_colsum_t')
    else:
        native.check(native.lib().dsgcn_colsum(_ptr(t), R, C, _ptr(out), _stream()), 'dsgcn_colsum')
    return out


def colsum(t, split_last=False):
    """Sum over dim 0 of a contiguous fp32 tensor (R, ...) -> (...), fp64 accumulation.  The kernel gives 32 columns
    to a block, so a tall, narrow input (e.g. 32768 x 25) would run on one CU: it is folded first — rows grouped g at
    a time into a (R/g, g*C) view, summed, and the g partial rows summed by a second tiny launch.
    split_last: input (R, C, k) -> output (k, C), each of the k sums a contiguous vector (no strided slices
    downstream, which would cost their consumers a copy each)."""
    R = t.shape[0]
    shape = t.shape[1:]
    C = t.numel() // max(R, 1)
    g = _fold(R, C)
    if g > 1:
        t = _colsum_raw(t, R // g, g * C)
        R = g
    if split_last:
        k = shape[-1]
        return _colsum_raw(t, R, C, k).view(k, C // k)
    return _colsum_raw(t, R, C).view(shape)


def _fold(R, C):
    """Row groups for the two-stage column sum.  Up to 512 rows one launch does it (a block's 32 row slices read 8 rows
    per batch, all loads of a batch in flight); taller inputs (per-plane partials: n*C rows) are dealt to g groups so
    that every first-stage block sees ~128 rows, and a second launch adds the g partial rows."""
    if R <= 512:
        return 1
    g = 1
    while R % (2 * g) == 0 and 2 * g * C < 16384 and R // g > 128:
        g *= 2
    return g


def colsum_pair(ta, tb, split_last_b=False):
    """colsum(ta), colsum(tb, split_last=split_last_b) with both reductions sharing their launches (two instead of up
    to four: the folded first stages together, then the second stages)."""
    dev = ta.device
    Ra, Rb = ta.shape[0], tb.shape[0]
    Ca, Cb = ta.numel() // Ra, tb.numel() // Rb
    kb = tb.shape[-1] if split_last_b else 1
    ga, gb = _fold(Ra, Ca), _fold(Rb, Cb)
    lib = native.lib()
    outa = torch.empty(Ca, device=dev, dtype=torch.float32)
    outb = torch.empty(Cb, device=dev, dtype=torch.float32)
    if ga > 1 and gb > 1:
        ma = torch.empty(ga * Ca, device=dev, dtype=torch.float32)
        mb = torch.empty(gb * Cb, device=dev, dtype=torch.float32)
        native.check(lib.dsgcn_colsum2(_ptr(ta), Ra // ga, ga * Ca, 1, _ptr(ma), _ptr(tb), Rb // gb, gb * Cb, 1, _ptr(mb),
                                       _stream()), 'dsgcn_colsum2')
        native.check(lib.dsgcn_colsum2(_ptr(ma), ga, Ca, 1, _ptr(outa), _ptr(mb), gb, Cb, kb, _ptr(outb), _stream()),
                     'dsgcn_colsum2')
    elif ga == 1 and gb == 1:
        native.check(lib.dsgcn_colsum2(_ptr(ta), Ra, Ca, 1, _ptr(outa), _ptr(tb), Rb, Cb, kb, _ptr(outb), _stream()),
                     'dsgcn_colsum2')
    elif ga == 1:
        # only the second input is tall (the GEMM-form convs write one partial row per 128-position tile): its first
        # stage rides with the whole first reduction, then its second stage alone — two launches, not three
        mb = torch.empty(gb * Cb, device=dev, dtype=torch.float32)
        native.check(lib.dsgcn_colsum2(_ptr(ta), Ra, Ca, 1, _ptr(outa), _ptr(tb), Rb // gb, gb * Cb, 1, _ptr(mb),
                                       _stream()), 'dsgcn_colsum2')
        outb = _colsum_raw(mb, gb, Cb, kb)
    else:
        ma = torch.empty(ga * Ca, device=dev, dtype=torch.float32)
        native.check(lib.dsgcn_colsum2(_ptr(ta), Ra // ga, ga * Ca, 1, _ptr(ma), _ptr(tb), Rb, Cb, kb, _ptr(outb),
                                       _stream()), 'dsgcn_colsum2')
        outa = _colsum_raw(ma, ga, Ca)
    return outa.view(ta.shape[1:]), (outb.view(kb, Cb // kb) if split_last_b else outb.view(tb.shape[1:]))


# ---------------------------------------------------------------------------------------------
# deferred column sums of parameter-gradient partials
# ---------------------------------------------------------------------------------------------
# The partial rows of weight / bias / A / alpha / beta / add_coeff gradients feed nothing but the optimizer.  Inside a
# ``deferred_param_sums()`` region (TrainEngine's backward) their column sums are queued instead of launched — ~55 launches
# of 5-6 us each per DS-STGCN step — and ``flush_param_sums()`` finishes all of them with ONE dsgcn_colsum_multi launch
# before the gradients are packed.  The queued outputs are handed to autograd UNFILLED, so a backward may only queue when
# every node between its parameter inputs and the leaves is a pure view (``_leafish``): nothing reads them before the flush.

_deferred = None
_VIEW_NODES = ('ViewBackward0', 'ReshapeAliasBackward0', 'UnsafeViewBackward0', 'AliasBackward0', '_CatRowsBackward')
_leaf_uses = {}          # id(leaf parameter) -> calls that asked to defer its gradient since the last deferred backward


class _LeafUse:
    """Result of ``_leafish``: truthy when every tensor reaches its leaves through view-only nodes AND — asked at BACKWARD
    time, when every forward of the step has registered — no leaf was registered by a second call.  A parameter shared by two
    deferring calls would have autograd ADD two unfilled gradients before the flush; those calls fall back to the
    immediate column sum instead (``dggcn`` shares ``A`` and says so with ``single_use=False``; this catches the ones that
    do not say)."""
    __slots__ = ('ok', 'leaves')

    def __init__(self, ok, leaves):
        self.ok, self.leaves = ok, tuple(leaves)
        if ok:
            for i in self.leaves:
                _leaf_uses[i] = _leaf_uses.get(i, 0) + 1

    def __bool__(self):
        return self.ok and all(_leaf_uses.get(i, 1) <= 1 for i in self.leaves)


def _leafish(*tensors):
    """Truthy (a ``_LeafUse``) when every tensor is a leaf or reaches its leaves through view-only autograd nodes."""
    leaves = []

    def ok(fn):
        for _ in range(8):
            if fn is None:
                return True
            if type(fn).__name__ == 'AccumulateGrad':
                leaves.append(id(fn.variable))
                return True
            if type(fn).__name__ not in _VIEW_NODES:
                return False
            nxt = [f for f, _ in fn.next_functions if f is not None]
            if len(nxt) != 1:
                return all(ok(f) for f in nxt)
            fn = nxt[0]
        return False

    good = True
    for t in tensors:
        if t is None or not t.requires_grad:
            continue
        if t.grad_fn is None:
            leaves.append(id(t))
        elif not ok(t.grad_fn):
            good = False
            break
    return _LeafUse(good, leaves)


def reset_leaf_uses():
    """Start of a step (TrainEngine._fwd_bwd): forget which leaves the previous step's forward registered."""
    _leaf_uses.clear()
    _coef_wait.clear()                # (jobs a failed backward left waiting)
    _wsplit_state['epoch'] += 1
    _wsplit_state['in_step'] = True


def end_step():
    """End of a step's forward + backward (TrainEngine._fwd_bwd): from here on the optimizer may rewrite the weights
    through raw pointers (dsgcn_sgd_step on the flat buffer: no version counter moves), so no cached weight image is
    trusted until the next reset_leaf_uses()."""
    _wsplit_state['in_step'] = False


# ---- pre-split weight images of the wide 1x1 convs: one launch per step ------------------------------------------------
# A wide conv's forward and data gradient read the three bf16 terms of W / W^T from an image that is rebuilt from the
# weights every step (csrc/pw4.hip k_wsplit): 22 dependent ~4 us launches per DS-STGCN step, each at the head of its conv.
# The images live here, keyed by (weight address, shape), and the first wide conv of a step (a step = reset_leaf_uses(),
# i.e. TrainEngine) rebuilds ALL known images with one dsgcn_pwconv_wsplit_multi launch.  An image is trusted only INSIDE
# a step (between reset_leaf_uses() and end_step(): the optimizer writes the weights through raw pointers, which moves no
# version counter) and only while the step AND the weight tensor's version counter are the ones it was built at; anything
# else (a conv not seen before, a forward outside a TrainEngine step — validation, inference —, weights changed in between)
# is split on the spot.  Images that were live while a hipGraph was being captured are pinned: the graph holds their
# addresses, so they are never pruned.
WSPLIT_BATCH = _os.environ.get('DSGCN_WSPLIT_BATCH', '1') == '1'
_wsplit_state = dict(epoch=0, batched=-1, jobs={}, in_step=False)


def _wsplit_image(w2, Ci, Co, nbytes):
    st = _wsplit_state
    jobs = st['jobs']
    key = (w2.data_ptr(), Ci, Co, nbytes, w2.device.index)
    job = jobs.get(key)
    capturing = w2.is_cuda and torch.cuda.is_current_stream_capturing()
    if job is None:
        if len(jobs) > 512:
            for k in [k for k, j in jobs.items() if not j['pinned']]:
                del jobs[k]
        job = jobs[key] = dict(w=w2, img=torch.empty(nbytes, device=w2.device, dtype=torch.uint8), stamp=None, used=0,
                               pinned=False)
    else:
        job['w'] = w2
    if capturing:
        job['pinned'] = True
    stamp = (st['epoch'], w2._version)
    if st['in_step'] and job['stamp'] == stamp:
        job['used'] = st['epoch']
        return job['img']
    lib = native.lib()
    if WSPLIT_BATCH and st['in_step'] and st['batched'] != st['epoch'] and job['stamp'] is not None:
        st['batched'] = st['epoch']
        # the convs of the previous step are the ones this step will run; images nobody asked for since (another model's,
        # a shape no longer in use) are dropped — unless a captured graph holds their address
        for k in [k for k, j in jobs.items() if j['used'] < st['epoch'] - 1 and j is not job and not j['pinned']]:
            del jobs[k]
        keys = [k for k, j in jobs.items() if k[4] == key[4] and j['stamp'] is not None]
        todo = [jobs[k] for k in keys]
        ws = (_ct.c_void_p * len(todo))(*[j['w'].data_ptr() for j in todo])
        out = (_ct.c_void_p * len(todo))(*[j['img'].data_ptr() for j in todo])
        native.check(lib.dsgcn_pwconv_wsplit_multi(ws, out, _int_array([k[1] for k in keys]), _int_array([k[2] for k in keys]),
                                                   len(todo), _stream()), 'dsgcn_pwconv_wsplit_multi')
        for j in todo:
            j['stamp'] = (st['epoch'], j['w']._version)
            if capturing:
                j['pinned'] = True
        if job['stamp'] == stamp:
            job['used'] = st['epoch']
            return job['img']
    native.check(lib.dsgcn_pwconv_wsplit(_ptr(w2), Ci, Co, _ptr(job['img']), _stream()), 'dsgcn_pwconv_wsplit')
    job['stamp'], job['used'] = stamp, st['epoch']
    return job['img']


class _StepBuilt:
    """Buffers derived from PARAMETERS only (CTR-GCN's augmented conv4 operands, the per-tap weight images of the dense
    temporal conv), rebuilt once per step: the first asker of a step rebuilds everything the previous step asked for with
    ONE launch.  The trust rule is _wsplit_image's: a cached buffer is used only inside the step it was built in
    (reset_leaf_uses() .. end_step(): the optimizer rewrites the weights through raw pointers) and only while the version
    counters of its source tensors are the ones it was built at; outside a step, or with the switch off, every call builds
    its own.  What the cache keeps is detached: a kept `conv.weight.view(...)` would keep last step's AccumulateGrad nodes —
    and the stream they were made on — alive into a graph capture (detach() shares storage and version counter).
    launch(jobs): one launch building every job's buffers; a job is a dict with 'src' (the source tensors, None allowed)
    plus whatever `alloc()` returned."""

    def __init__(self, launch, enabled, limit=256):
        self.launch, self.enabled, self.limit = launch, enabled, limit
        self.jobs, self.batched = {}, -1

    def clear(self):
        self.jobs, self.batched = {}, -1

    @staticmethod
    def _versions(src):
        return tuple(-1 if t is None else t._version for t in src)

    def get(self, shape_key, src, alloc):
        ws = _wsplit_state
        src = [None if t is None else t.detach() for t in src]
        if not (self.enabled() and ws['in_step']):
            job = dict(src=src, **alloc())
            self.launch([job])
            return job
        dev = next(t for t in src if t is not None).device.index
        key = (tuple(0 if t is None else t.data_ptr() for t in src), shape_key, dev)
        capturing = torch.cuda.is_current_stream_capturing()
        jobs = self.jobs
        job = jobs.get(key)
        if job is None:
            if len(jobs) > self.limit:
                for k in [k for k, j in jobs.items() if not j['pinned']]:
                    del jobs[k]
            job = jobs[key] = dict(stamp=None, used=0, pinned=False, **alloc())
        job['src'] = src
        if capturing:
            job['pinned'] = True                  # the graph holds the buffers' addresses: never pruned
        stamp = (ws['epoch'], self._versions(src))
        job['used'] = ws['epoch']
        if job['stamp'] == stamp:
            return job
        if self.batched != ws['epoch'] and job['stamp'] is not None:
            self.batched = ws['epoch']
            for k in [k for k, j in jobs.items() if j['used'] < ws['epoch'] - 1 and j is not job and not j['pinned']]:
                del jobs[k]
            todo = [j for k, j in jobs.items() if k[2] == dev and j['stamp'] is not None]
            self.launch(todo)
            for j in todo:
                j['stamp'] = (ws['epoch'], self._versions(j['src']))
                if capturing:
                    j['pinned'] = True
            if job['stamp'] == stamp:
                return job
        self.launch([job])
        job['stamp'] = stamp
        return job


class deferred_param_sums:
    """``with deferred_param_sums(flat): loss.backward()`` — see above; flushes on exit.  The queued outputs reach autograd
    unfilled, so the region insists on what makes that safe: ``FlatParams(gather=True)`` (no in-place accumulation into
    pre-assigned ``.grad`` views) and no ``.grad`` present on entry (AccumulateGrad would add the unfilled tensor to it)."""

    def __init__(self, flat=None):
        self.flat = flat

    def __enter__(self):
        global _deferred
        if self.flat is not None:
            if not self.flat.gather:
                raise RuntimeError('deferred_param_sums needs FlatParams(gather=True): with gradient views autograd '
                                   'accumulates in place, before the deferred sums are filled')
            if any(p.grad is not None for p in self.flat.params):
                raise RuntimeError('deferred_param_sums: a parameter already holds a .grad — call zero_grad() first '
                                   '(gradient accumulation over several backward passes is not supported here)')
        self.prev = _deferred
        _deferred = []
        return self

    def __exit__(self, *exc):
        global _deferred
        try:
            if exc[0] is None:
                flush_param_sums()
        finally:
            _deferred = self.prev
            _leaf_uses.clear()
            _post_flush.clear()                     # (a failed backward must not leave finishing launches behind)
            _ctr_fin_queue.clear()
        return False


_post_flush = []         # finishing launches that read deferred sums (run by flush_param_sums after the last level)


def flush_param_sums():
    """One launch per level for every queued column sum, then the finishing launches registered in ``_post_flush``
    (no-op when nothing is queued)."""
    global _deferred
    queued = _deferred
    if queued:
        _deferred = [] if _deferred is not None else None
        for level in sorted({j[4] for j in queued}):  # level 1 = second stages of tall inputs: read what level 0 wrote
            _flush_jobs([j[:4] for j in queued if j[4] == level])
    hooks = list(_post_flush)
    _post_flush.clear()
    for h in hooks:
        h()


def _flush_jobs(jobs):
    """One dsgcn_colsum_multi_host call: the job table rides in the kernel arguments (no device table, no upload — the
    pinned staging slots, their events and the copy launch of the first version are gone)."""
    import numpy as _np
    k = len(jobs)
    tab = _np.empty((k, 4), dtype=_np.int64)
    blk = 0
    nblocks = native.lib().dsgcn_colsum_blocks
    for i, (src, R, C, out) in enumerate(jobs):
        tab[i] = (src.data_ptr(), out.data_ptr(), (R << 32) | C, blk)
        blk += nblocks(src.data_ptr(), C)
    native.check(native.lib().dsgcn_colsum_multi_host(tab.ctypes.data, k, _stream()), 'dsgcn_colsum_multi_host')


def param_colsum(t, defer_ok=True, level=0):
    """colsum(t) for partial rows that feed only parameter gradients: queued inside a deferred_param_sums() region (the
    result is then filled by flush_param_sums()), immediate otherwise.  level: 1 for an input that is itself the (still
    unfilled) result of a level-0 call — the flush runs its launches level by level."""
    if _deferred is None or not defer_ok or not t.is_cuda:
        return colsum(t)
    R = t.shape[0]
    shape = t.shape[1:]
    C = t.numel() // max(R, 1)
    g = _fold(R, C)
    out = torch.empty(C, device=t.device, dtype=torch.float32)
    if g > 1:                                         # tall inputs: the wide first stage in the flush's first launch, the
        mid = torch.empty(g * C, device=t.device, dtype=torch.float32)         # small second one in its second
        _deferred.append((t, R // g, g * C, mid, level))
        _deferred.append((mid, g, C, out, level + 1))
    else:
        _deferred.append((t, R, C, out, level))
    return out.view(shape)


class _CatRows(torch.autograd.Function):
    """torch.cat(tensors, 0) for parameter tensors.  When the tensors already lie back to back in one storage (FlatParams
    lays the groups its modules declare in ``flat_groups()`` out that way) the result is a view of that storage — no
    launch; the gradient is handed back as row slices either way (what CatBackward does)."""

    @staticmethod
    def forward(ctx, *ts):
        ctx.rows = [t.shape[0] for t in ts]
        t0 = ts[0]
        adjacent = all(t.is_contiguous() and t.dtype == t0.dtype and t.device == t0.device and
                       t.shape[1:] == t0.shape[1:] for t in ts)
        if adjacent:
            esz = t0.element_size()
            base = t0.untyped_storage().data_ptr()
            for a, b in zip(ts[:-1], ts[1:]):
                if b.untyped_storage().data_ptr() != base or a.data_ptr() + a.numel() * esz != b.data_ptr():
                    adjacent = False
                    break
        if not adjacent:
            return torch.cat(ts, 0)
        size = (sum(ctx.rows),) + tuple(t0.shape[1:])
        stride = []
        acc = 1
        for d in reversed(size):
            stride.append(acc)
            acc *= d
        return torch.as_strided(t0.detach(), size, tuple(reversed(stride)))

    @staticmethod
    def backward(ctx, g):
        return tuple(g.split(ctx.rows, 0))


def cat_rows(tensors):
    tensors = list(tensors)
    return tensors[0] if len(tensors) == 1 else _CatRows.apply(*tensors)


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise TypeError(f'dsgcn kernels compute in fp32, got {t.dtype}')
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------
# Deferred BatchNorm backward: the consumer hands the producer finished coefficients
# ---------------------------------------------------------------------------------------------
# A deferred BatchNorm's (scale, shift) is consumed by the next kernel's load prologue; in backward that consumer holds
# partial rows of d scale / d shift and the producer needs (A0, B0, dgamma, dbeta) from their column sums.  Through
# autograd alone that is a column-sum launch in the consumer plus a coefficient launch in the producer (55 + 55 per
# DS-STGCN step, 4-5 us each).  BNCtx travels with the scale tensor (a Python attribute set where the producer returns
# it): a consumer that finds one writes the coefficients itself with ONE launch (dsgcn_bn_coef_rows; a second consumer of
# the same BatchNorm adds to them — the coefficients are linear in the sums) and returns no gradient for scale / shift;
# autograd still runs the producer's backward after all of its consumers'.  Consumers without the hook (torch ops, the
# units this has not been wired into) return ordinary gradients and the producer adds their share the old way.
BN_FEED = _os.environ.get('DSGCN_BN_FEED', '1') != '0'


class BNCtx:
    __slots__ = ('mean', 'var', 'gamma', 'eps', 'count', 'C', 'n_affine', 'coef', 'host', 'queued')

    def __init__(self):
        self.coef = None
        self.host = False           # its backward coefficients may wait for a hosting launch (K-B's backward)
        self.queued = None          # the coefficient job that waits


def _bn_attach(scale, bn, mean, var, gamma, eps, count, n_affine):
    """producer side (outside its autograd Function): make the returned scale tensor carry the BatchNorm's context"""
    if bn is None or scale is None or not BN_FEED:
        return
    bn.mean, bn.var, bn.gamma, bn.eps, bn.count = mean, var, gamma, float(eps), float(count)
    bn.C, bn.n_affine = int(scale.numel()), int(n_affine)
    scale._dsgcn_bn = bn


def _bn_of(scale):
    return getattr(scale, '_dsgcn_bn', None) if (scale is not None and BN_FEED) else None


# ---- BatchNorm micro-launches as jobs (csrc/bn_jobs.h; round 6) ---------------------------------------------------------
# A step had ~110 dependent 4-5 us launches of k_bn_finalize / k_bn_coef_rows.  A job = the arguments of one of them; jobs
# whose producers are independent of one another go out in ONE launch (`post` + `down`, transform + the block's residual
# conv), and the `pre` BatchNorm's jobs ride as extra workgroups of K-B's launches, which sit between the `pre` conv and K-A
# anyway.  Same blocks, same summation order: bit-identical to the single launches.  DSGCN_BN_BATCH=0 restores them.
BN_BATCH = _os.environ.get('DSGCN_BN_BATCH', '1') == '1'
_fin_queue = None            # list while a bn_batch() region is open: finalize jobs wait there
_coef_wait = []              # coefficient jobs that wait for a hosting launch (see BNCtx.host)


def _fin_job(partial, rows, Co, count, gamma, beta, eps, mean, var, scale, shift, n_affine):
    j = native.BnFinJob(_ptr(partial), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(var), _ptr(scale), _ptr(shift), float(count),
                        float(eps), int(rows), int(Co), int(n_affine))
    j._keep = (partial, gamma, beta, mean, var, scale, shift)
    return j


def _launch_fin(jobs):
    lib = native.lib()
    for i in range(0, len(jobs), native.BN_JOBS_MAX):
        chunk = jobs[i:i + native.BN_JOBS_MAX]
        arr = (native.BnFinJob * len(chunk))(*chunk)
        native.check(lib.dsgcn_bn_finalize_multi(arr, len(chunk), _stream()), 'dsgcn_bn_finalize_multi')


def bn_finalize(partial, rows, Co, count, gamma, beta, eps, mean, var, scale, shift, n_affine):
    """partial rows -> (mean, var, scale, shift): queued inside a bn_batch() region, launched otherwise."""
    if _fin_queue is not None:
        _fin_queue.append(_fin_job(partial, rows, Co, count, gamma, beta, eps, mean, var, scale, shift, n_affine))
        return
    rc = native.lib().dsgcn_bn_finalize(_ptr(partial), rows, Co, count, _ptr(gamma), _ptr(beta), float(eps), _ptr(mean),
                                        _ptr(var), _ptr(scale), _ptr(shift), int(n_affine), _stream())
    native.check(rc, 'dsgcn_bn_finalize')


class bn_batch:
    """``with bn_batch() as q:`` — the BatchNorm finalizes of the convs launched inside wait in ``q`` and go out as ONE launch
    when the region closes (or as extra workgroups of a launch that takes them: ``dynadj(..., host=q)``).  ``with q.hold():``
    is a region that closes WITHOUT launching: the jobs wait for a later ``with q:`` (the block's residual conv is issued
    first, its finalize joins the transform conv's).  NOTHING between a conv and the launch of its job may read that conv's
    (scale, shift): the caller states the independence by writing the region."""

    def __init__(self):
        self.jobs = []
        self._stack = []

    def _enter(self, flush):
        global _fin_queue
        self._stack.append((_fin_queue, flush))
        if BN_BATCH:
            _fin_queue = self.jobs
        return self

    def __enter__(self):
        return self._enter(True)

    def hold(self):
        outer = self

        class _Hold:
            def __enter__(self):
                return outer._enter(False)

            def __exit__(self, *exc):
                return outer.__exit__(*exc)
        return _Hold()

    def take(self):
        jobs, self.jobs[:] = list(self.jobs), []
        return jobs

    def flush(self):
        jobs = self.take()
        if jobs:
            _launch_fin(jobs)

    def __exit__(self, *exc):
        global _fin_queue
        _fin_queue, flush = self._stack.pop()
        if exc[0] is not None:
            self.jobs[:] = []
        elif flush:
            self.flush()
        return False


def _coef_job(bn, part, k, i_ds, i_dh):
    C = bn.C
    R = part.numel() // (C * k)
    acc = bn.coef is not None
    if not acc:
        bn.coef = torch.empty((4, C), device=part.device, dtype=torch.float32)
    j = native.BnCoefJob(_ptr(part), _ptr(bn.mean), _ptr(bn.var), _ptr(bn.gamma), _ptr(bn.coef), float(bn.count),
                         float(bn.eps), int(R), int(C), int(k), int(i_ds), int(i_dh), int(bn.n_affine), int(acc))
    j._keep = (part, bn.mean, bn.var, bn.gamma, bn.coef)
    return j


def _launch_coef(jobs):
    lib = native.lib()
    for i in range(0, len(jobs), native.BN_JOBS_MAX):
        chunk = jobs[i:i + native.BN_JOBS_MAX]
        arr = (native.BnCoefJob * len(chunk))(*chunk)
        native.check(lib.dsgcn_bn_coef_rows_multi(arr, len(chunk), _stream()), 'dsgcn_bn_coef_rows_multi')


def _flush_coef_wait():
    if _coef_wait:
        jobs = [j for _, j in _coef_wait]
        for bn, _ in _coef_wait:
            bn.queued = None
        _coef_wait.clear()
        _launch_coef(jobs)


def _bn_feed(bn, part, k, i_ds, i_dh):
    """consumer side, in backward: partial rows part (R, C, k) -> the producer's coefficients (one launch; for a BatchNorm
    marked ``host`` the job waits for K-B's backward launch, or for the producer's own backward, whichever comes first)."""
    _bn_feed_multi([(bn, part, k, i_ds, i_dh)])


def _job_array(jobs):
    """-> (ctypes array | None, count) for the `jobs, njobs` arguments of a hosting entry point"""
    return ((native.BnCoefJob * len(jobs))(*jobs) if jobs else None), len(jobs)


def _bn_feed_multi(feeds, launch=True):
    """several consumers' rows in one launch (the two BatchNorms of a two-stream operand: `post` + `down`, ...).
    launch=False: -> the jobs, for a caller whose NEXT launch hosts them (a weight gradient: nothing on the critical chain
    waits for it, and the rows come from the data gradient before it) — [] when batching is off (launched here then)."""
    if not BN_BATCH:
        for bn, part, k, i_ds, i_dh in feeds:
            C = bn.C
            R = part.numel() // (C * k)
            acc = bn.coef is not None
            if not acc:
                bn.coef = torch.empty((4, C), device=part.device, dtype=torch.float32)
            rc = native.lib().dsgcn_bn_coef_rows(_ptr(part), R, C, k, i_ds, i_dh, _ptr(bn.mean), _ptr(bn.var), _ptr(bn.gamma),
                                                 bn.eps, bn.count, bn.n_affine, _ptr(bn.coef), int(acc), _stream())
            native.check(rc, 'dsgcn_bn_coef_rows')
        return []
    now = []
    for bn, part, k, i_ds, i_dh in feeds:
        if bn.queued is not None:            # a second consumer adds to what the first one's job writes: keep the order
            _flush_coef_wait()
        job = _coef_job(bn, part, k, i_ds, i_dh)
        if bn.host:
            bn.queued = job
            _coef_wait.append((bn, job))
        else:
            now.append(job)
    if not launch and len(now) <= native.BN_JOBS_MAX:
        return now
    if now:
        _launch_coef(now)
    return []


def _bn_coef(bn, gscale, gshift, mean, var, gamma, eps, count, C, n_affine):
    """producer side, in backward: -> (dgamma, dbeta, A0, B0) or four Nones when nothing reached the BatchNorm."""
    coef = None
    if bn is not None:
        if bn.queued is not None:           # no hosting launch came by: the job goes out now
            _flush_coef_wait()
        coef, bn.coef = bn.coef, None
    if gscale is not None or gshift is not None:
        c2 = torch.empty((4, C), device=mean.device, dtype=torch.float32)
        rc = native.lib().dsgcn_bn_bwd_coef(_ptr(gscale), _ptr(gshift), _ptr(mean), _ptr(var), _ptr(gamma), eps, count, C,
                                            n_affine, _ptr(c2[0]), _ptr(c2[1]), _ptr(c2[2]), _ptr(c2[3]), _stream())
        native.check(rc, 'dsgcn_bn_bwd_coef')
        coef = c2 if coef is None else coef + c2
    if coef is None:
        return None, None, None, None
    return coef[0], coef[1], coef[2], coef[3]


# ---------------------------------------------------------------------------------------------
# K-A  gather-aggregate
# ---------------------------------------------------------------------------------------------

class _Aggregate(torch.autograd.Function):

    @staticmethod
    def forward(ctx, zp, scale, shift, relu, ahat):
        _require_cuda(zp, ahat)
        zp, ahat, scale, shift = _f32c(zp), _f32c(ahat), _f32c(scale), _f32c(shift)
        n, KC, T, V = zp.shape
        assert ahat.shape == (n, KC, V, V), (ahat.shape, zp.shape)
        y = torch.empty_like(zp)
        rc = native.lib().dsgcn_aggregate_fwd(_ptr(zp), _ptr(scale), _ptr(shift), int(relu), _ptr(ahat), _ptr(y),
                                              n, KC, T, V, _stream())
        native.check(rc, 'dsgcn_aggregate_fwd')
        ctx.save_for_backward(zp, scale, shift, ahat)
        ctx.relu = int(relu)
        ctx.bn = _bn_of(scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        zp, scale, shift, ahat = ctx.saved_tensors
        n, KC, T, V = zp.shape
        dy = _f32c(dy)
        dzp = torch.empty_like(zp)
        dahat = torch.empty_like(ahat)
        rows = native.lib().dsgcn_aggregate_bwd_partial_rows(n, T, V)
        partial = torch.empty((rows, KC, 2), device=zp.device, dtype=torch.float32)
        rc = native.lib().dsgcn_aggregate_bwd(_ptr(zp), _ptr(scale), _ptr(shift), ctx.relu, _ptr(ahat), _ptr(dy),
                                              _ptr(dzp), _ptr(dahat), _ptr(partial), n, KC, T, V, _stream())
        native.check(rc, 'dsgcn_aggregate_bwd')
        dscale = dshift = None
        if scale is not None and ctx.bn is not None:
            _bn_feed(ctx.bn, partial, 2, 0, 1)
        elif scale is not None:
            red = colsum(partial, split_last=True)
            dscale, dshift = red[0], red[1]
        return dzp, dscale, dshift, None, dahat


def aggregate(zp, ap, relu, ahat):
    scale, shift = ap if ap is not None else (None, None)
    return _Aggregate.apply(zp, scale, shift, relu, ahat)


# ---------------------------------------------------------------------------------------------
# K-B  dynamic adjacency
# ---------------------------------------------------------------------------------------------

class _DynAdj(torch.autograd.Function):
    """proj (n, (4+P)*mid, ld) rows [a | b | s-typed] (ld >= V: padded joint stride) -> Ahat (n, 3*mid, V, V)
    (K-B, one HIP launch each way)."""

    @staticmethod
    def forward(ctx, proj, A, alpha, beta, we, be, node_type, edge_type, single_use=True, host=None):
        """host: a ``bn_batch`` whose waiting finalize jobs ride in this launch as extra workgroups."""
        _require_cuda(proj, A)
        proj, A, alpha, beta, we, be = [_f32c(t) for t in (proj, A, alpha, beta, we, be)]
        n, R, ld = proj.shape
        V = A.shape[-1]
        mid = we.shape[1]
        E = we.shape[0] // mid
        P = (R // mid - 4)
        assert A.shape[0] == 3 and node_type.dtype == torch.int32 and edge_type.dtype == torch.int32
        ahat = torch.empty((n, 3 * mid, V, V), device=proj.device, dtype=torch.float32)
        jobs = host.take() if host is not None else []
        if len(jobs) > native.BN_JOBS_MAX:
            _launch_fin(jobs)
            jobs = []
        arr = (native.BnFinJob * len(jobs))(*jobs) if jobs else None
        rc = native.lib().dsgcn_dynadj_fwd_jobs(_ptr(proj), _ptr(A), _ptr(alpha), _ptr(beta), _ptr(we), _ptr(be),
                                                _ptr(node_type), _ptr(edge_type), _ptr(ahat), n, mid, V, ld, P, E, arr,
                                                len(jobs), _stream())
        native.check(rc, 'dsgcn_dynadj_fwd_jobs')
        ctx.save_for_backward(proj, alpha, beta, we, be, node_type, edge_type)
        ctx.dims = (n, mid, V, ld, P, E)
        ctx.defer_ok = bool(single_use) and _leafish(A, alpha, beta, we, be)
        return ahat

    @staticmethod
    def backward(ctx, dahat):
        proj, alpha, beta, we, be, node_type, edge_type = ctx.saved_tensors
        n, mid, V, ld, P, E = ctx.dims
        dahat = _f32c(dahat)
        dev = proj.device
        lib = native.lib()
        dd = torch.empty_like(dahat)
        dproj = torch.empty_like(proj)
        pstride = lib.dsgcn_dynadj_partial_stride(mid, V, E)
        ppar = torch.empty((n, pstride), device=dev, dtype=torch.float32)      # per-sample parameter-gradient partials
        # coefficient jobs that wait for a hosting launch (the `pre` BatchNorm's, queued by K-A's backward) ride here
        waiting = _coef_wait[:native.BN_JOBS_MAX]
        del _coef_wait[:len(waiting)]
        for bn, _ in waiting:
            bn.queued = None
        jobs = [j for _, j in waiting]
        arr = (native.BnCoefJob * len(jobs))(*jobs) if jobs else None
        rc = lib.dsgcn_dynadj_bwd_jobs(_ptr(proj), _ptr(alpha), _ptr(beta), _ptr(we), _ptr(be), _ptr(node_type),
                                       _ptr(edge_type), _ptr(dahat), _ptr(dd), _ptr(dproj), _ptr(ppar), pstride, n, mid, V,
                                       ld, P, E, arr, len(jobs), _stream())
        native.check(rc, 'dsgcn_dynadj_bwd_jobs')
        red = param_colsum(ppar, ctx.defer_ok)                                  # ordered sum over samples: deterministic
        o = 3 * V * V
        dA, dalpha, dbeta = red[:o].view(3, V, V), red[o:o + 3], red[o + 3:o + 6]
        dwe = red[o + 6:o + 6 + E * mid * mid].view(E * mid, mid)
        dbe = red[o + 6 + E * mid * mid:o + 6 + E * mid * mid + E * mid]
        return dproj, dA, dalpha, dbeta, dwe, dbe, None, None, None, None


def dynadj(xbar, A, alpha, beta, w1, b1, w2, b2, wse, bse, we, be, node_type, edge_type, single_use=True, host=None):
    """Dynamic adjacency.  The three mean-pooled projections (conv1/conv2/conv1_se) are one K-C launch on xbar — viewed
    as a (n, Ci, 1, 32) "clip" with the joint rows zero-padded to 32, so that forward, data gradient and weight gradient
    all take the 16-byte-per-lane K-C kernels (an unpadded 25-joint row is odd-sized: it fell to the scalar-load kernels,
    ~35 us per launch for 0.1 GFLOP) — and the rest is K-B reading / writing the padded rows."""
    n, Ci, _ = xbar.shape
    V = A.shape[-1]
    w_all = cat_rows([w1, w2, wse])
    b_all = cat_rows([b1, b2, bse])
    # xbar arrives zero-padded to 32 joints from the previous block's fuse_out (want_tmean=32); the first block pads here
    xpad = torch.nn.functional.pad(xbar, (0, 32 - V)) if xbar.shape[-1] < 32 else xbar
    proj = pwconv(xpad.unsqueeze(2), None, None, None, False, w_all, b_all, 1, False)[0]
    # single_use: every parameter passed here is used by this call only in the step (their gradient partials may then join
    # the end-of-backward sum, see param_colsum); dggcn feeds A to two calls and says so
    return _DynAdj.apply(proj.view(n, w_all.shape[0], xpad.shape[-1]), A, alpha, beta, we, be, node_type, edge_type,
                         bool(single_use), host)


# ---------------------------------------------------------------------------------------------
# K-C  1x1 channel mix + train-mode BN statistics / deferred affine
# ---------------------------------------------------------------------------------------------

import ctypes as _ct


def _pw_plan(Tout, V, aug):
    tr, npad, nb = _ct.c_int(), _ct.c_int(), _ct.c_int()
    native.check(native.lib().dsgcn_pwconv_plan(Tout, V, int(aug), _ct.byref(tr), _ct.byref(npad), _ct.byref(nb)),
                 'dsgcn_pwconv_plan')
    return tr.value, npad.value, nb.value


class _PwConv(torch.autograd.Function):
    """z = W . virt(x1,a1,x2,a2,relu)[::stride] + b ; optional zaug = mean_v z ; optional BN of z:
    (scale, shift) = (gamma*rsqrt(var+eps), beta-mean*scale) from the batch statistics of z (+zaug)."""

    @staticmethod
    def forward(ctx, x1, s1, h1, x2, s2, h2, relu, weight, bias, stride, aug, gamma, beta, eps, n_affine, want_bn,
                bn=None, out=None):
        _require_cuda(x1, weight)
        ctx.bn, ctx.bn1, ctx.bn2 = bn, _bn_of(s1), _bn_of(s2)
        x1, s1, h1, x2, s2, h2, bias, gamma, beta = [_f32c(t) for t in (x1, s1, h1, x2, s2, h2, bias, gamma, beta)]
        w2 = _f32c(weight.reshape(weight.shape[0], -1))
        n, Ci, T, V = x1.shape
        Co = w2.shape[0]
        assert w2.shape[1] == Ci, (w2.shape, x1.shape)
        Tout = (T + stride - 1) // stride
        dev = x1.device
        if out is not None:
            z = out.t
            if (tuple(z.shape) != (n, Co, Tout, V) or z.dtype != torch.float32 or not z.is_contiguous() or z.device != dev
                    or z.requires_grad):
                raise ValueError(f'pwconv(out=): expected a contiguous fp32 {(n, Co, Tout, V)} slice without history')
        else:
            z = torch.empty((n, Co, Tout, V), device=dev, dtype=torch.float32)
        zaug = torch.empty((n, Co, Tout), device=dev, dtype=torch.float32) if aug else None
        lib = native.lib()
        partial = None
        if want_bn:
            rows = lib.dsgcn_pwconv_partial_rows(n, Ci, Co, T, V, stride, int(aug))
            partial = torch.empty((rows, Co, 2), device=dev, dtype=torch.float32)
        # wide convs: the three bf16 terms of the weights, split once here for the forward and the data gradient
        ws = None
        wsb = lib.dsgcn_pwconv_wsplit_bytes(n, Ci, Co, T, V, stride)
        if wsb:
            ws = _wsplit_image(w2, Ci, Co, wsb)
        rc = lib.dsgcn_pwconv_fwd_ws(_ptr(x1), _ptr(s1), _ptr(h1), _ptr(x2), _ptr(s2), _ptr(h2), int(relu), _ptr(w2),
                                     _ptr(bias), _ptr(z), _ptr(zaug), _ptr(partial), n, Ci, Co, T, V, stride, int(aug),
                                     int(want_bn), _ptr(ws), _stream())
        native.check(rc, 'dsgcn_pwconv_fwd_ws')
        scale = shift = mean = var = None
        count = float(n * Tout * (V + (1 if aug else 0)))
        if want_bn:
            stats = torch.empty((4, Co), device=dev, dtype=torch.float32)
            mean, var, scale, shift = stats[0], stats[1], stats[2], stats[3]
            bn_finalize(partial, partial.shape[0], Co, count, gamma, beta, eps, mean, var, scale, shift, n_affine)
            ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)      # no zero tensors for the unused / non-differentiable outputs (mean, var)
        ctx.save_for_backward(x1, s1, h1, x2, s2, h2, w2, z, zaug, gamma, mean, var, ws)
        ctx.cfg = (int(relu), stride, int(aug), float(eps), int(n_affine), bool(want_bn), count, tuple(weight.shape),
                   bias is not None, beta is not None)
        ctx.defer_ok = _leafish(weight, bias)
        # the weight (and input scale) come from a producer that finishes their gradients after the deferred sums
        # (_CtrWPrep): both partial-row sums of this conv may then join the end-of-backward launches
        # The decision is the PRODUCER's _LeafUse object, asked again at backward time (when _CtrWPrep.backward asks it): one
        # answer for both sides — a conv4 leaf registered by two deferring calls makes both fall back to immediate sums.
        sink = getattr(weight, '_dsgcn_sink', None)
        if sink is not None and not (bias is None and s2 is None and (s1 is None or getattr(s1, '_dsgcn_sink', None) is sink)):
            sink = None
        ctx.sink = sink
        return z, zaug, scale, shift, mean, var

    @staticmethod
    def backward(ctx, gz, gzaug, gscale, gshift, _gm, _gv):
        x1, s1, h1, x2, s2, h2, w2, z, zaug, gamma, mean, var, ws = ctx.saved_tensors
        relu, stride, aug, eps, n_affine, want_bn, count, wshape, has_bias, has_beta = ctx.cfg
        n, Ci, T, V = x1.shape
        Co = w2.shape[0]
        dev = x1.device
        lib = native.lib()
        st = _stream()
        gz, gzaug, gscale, gshift = _f32c(gz), _f32c(gzaug), _f32c(gscale), _f32c(gshift)
        A0 = B0 = dgamma = dbeta = None
        if want_bn:
            dgamma, dbeta, A0, B0 = _bn_coef(ctx.bn, gscale, gshift, mean, var, gamma, eps, count, Co, n_affine)
        Tout = z.shape[2]
        if gz is None:
            # z itself received no gradient (only its statistics did): the kernels size their partial rows for the
            # wide-load path, which needs the operand — hand them explicit zeros instead of a NULL (the scalar
            # fallback taken for a NULL wrote more ipart rows than dsgcn_pwconv_ipart_rows reports)
            gz = torch.zeros_like(z)
        if aug:
            # fold the statistics terms and the global-joint gradient once; dgrad / wgrad then run in plain mode
            dzeff = torch.empty_like(z)
            rc = lib.dsgcn_dz_eff_aug(_ptr(gz), _ptr(z), _ptr(gzaug), _ptr(zaug), _ptr(A0), _ptr(B0), _ptr(dzeff), n,
                                      Co, Tout, V, st)
            native.check(rc, 'dsgcn_dz_eff_aug')
            gz, gzaug, A0, B0, aug = dzeff, None, None, None, 0
        dx1 = torch.empty_like(x1)
        dx2 = torch.empty_like(x2) if x2 is not None else None
        pstride = Co * Ci + Co
        rows = lib.dsgcn_pwconv_bwd_rows(n, Ci, Co, T, V, stride) if (not aug and gz is not None) else 0
        if rows > 0:
            # narrow conv: data gradient, weight gradient and the input-affine sums in one pass (csrc/bwd64.hip)
            wpart = torch.empty((rows, pstride), device=dev, dtype=torch.float32)
            ipart = (torch.empty((rows, Ci, 3), device=dev, dtype=torch.float32)
                     if (s1 is not None or s2 is not None) else None)
            rc = lib.dsgcn_pwconv_bwd(_ptr(x1), _ptr(s1), _ptr(h1), _ptr(x2), _ptr(s2), _ptr(h2), relu, _ptr(w2), _ptr(z),
                                      _ptr(gz), _ptr(A0), _ptr(B0), _ptr(dx1), _ptr(dx2), _ptr(ipart), wpart.data_ptr(),
                                      wpart.data_ptr() + 4 * Co * Ci, pstride, n, Ci, Co, T, V, st)
            native.check(rc, 'dsgcn_pwconv_bwd')
            return _PwConv._finish(wpart, ipart, dx1, dx2, s1, s2, Co, Ci, wshape, has_bias, dgamma, dbeta, gamma,
                                   has_beta, n_affine, ctx.defer_ok, ctx.bn1, ctx.bn2, ctx.sink is not None and bool(ctx.sink))
        ipart = None
        if s1 is not None or s2 is not None:
            rows = lib.dsgcn_pwconv_ipart_rows(n, Ci, Co, T, V, stride)
            ipart = torch.empty((rows, Ci, 3), device=dev, dtype=torch.float32)     # every row is written by dgrad
        rc = lib.dsgcn_pwconv_dgrad_ws(_ptr(x1), _ptr(s1), _ptr(h1), _ptr(x2), _ptr(s2), _ptr(h2), relu, _ptr(w2),
                                       _ptr(z), _ptr(zaug), _ptr(gz), _ptr(gzaug), _ptr(A0), _ptr(B0), _ptr(dx1),
                                       _ptr(dx2), _ptr(ipart), n, Ci, Co, T, V, stride, aug, _ptr(ws), st)
        native.check(rc, 'dsgcn_pwconv_dgrad_ws')
        # (the weight gradient on a second stream beside the data gradient was measured twice: round 2 17.65 vs 17.68
        # ms/step, round 3 13.33 vs 13.34 — it stays on the one stream)
        splits = lib.dsgcn_pwconv_wgrad_splits(n, Ci, Co, T, V, stride)
        pstride = Co * Ci + Co
        wpart = torch.empty((splits, pstride), device=dev, dtype=torch.float32)
        sink = ctx.sink is not None and bool(ctx.sink)
        # the input BatchNorms' coefficient jobs (rows: the data gradient above) ride in the weight gradient's launch
        hosted, fed = [], False
        if (ipart is not None and not (sink and _deferred is not None) and (s1 is None or ctx.bn1 is not None) and
                (s2 is None or ctx.bn2 is not None)):
            hosted = _bn_feed_multi(([(ctx.bn1, ipart, 3, 0, 1)] if s1 is not None else []) +
                                    ([(ctx.bn2, ipart, 3, 2, 1)] if s2 is not None else []), launch=False)
            fed = True
        rc = lib.dsgcn_pwconv_wgrad_jobs(_ptr(x1), _ptr(s1), _ptr(h1), _ptr(x2), _ptr(s2), _ptr(h2), relu, _ptr(z),
                                         _ptr(zaug), _ptr(gz), _ptr(gzaug), _ptr(A0), _ptr(B0), wpart.data_ptr(),
                                         wpart.data_ptr() + 4 * Co * Ci, pstride, n, Ci, Co, T, V, stride, aug,
                                         *_job_array(hosted), st)
        native.check(rc, 'dsgcn_pwconv_wgrad_jobs')
        return _PwConv._finish(wpart, ipart, dx1, dx2, s1, s2, Co, Ci, wshape, has_bias, dgamma, dbeta, gamma, has_beta,
                               n_affine, ctx.defer_ok, ctx.bn1, ctx.bn2, sink, fed)

    @staticmethod
    def _finish(wpart, ipart, dx1, dx2, s1, s2, Co, Ci, wshape, has_bias, dgamma, dbeta, gamma, has_beta, n_affine,
                defer_ok=False, bn1=None, bn2=None, sink=False, already_fed=False):
        """Ordered sums of the partial rows -> the gradient tuple of backward().  already_fed: the caller has handed the
        input BatchNorms' coefficient jobs to a hosting launch."""
        if sink and _deferred is not None:
            # both sums deferred; the input-scale gradient goes on as the strided column 0 of the summed (Ci, 3) rows
            wsum = param_colsum(wpart, True)
            ds1 = dh1 = None
            if ipart is not None:
                red = param_colsum(ipart.view(ipart.shape[0], -1), True).view(Ci, 3)
                ds1, dh1 = red[:, 0], red[:, 1]
            return (dx1, ds1, dh1, dx2, None, None, None, wsum[:Co * Ci].view(wshape), None, None, None, None, None, None,
                    None, None, None, None)
        fed = ipart is not None and (s1 is None or bn1 is not None) and (s2 is None or bn2 is not None)
        if fed and already_fed:
            ipart = None
        elif fed:
            # every affine of the virtual input belongs to a deferred BatchNorm that takes its coefficients directly
            _bn_feed_multi(([(bn1, ipart, 3, 0, 1)] if s1 is not None else []) + ([(bn2, ipart, 3, 2, 1)] if s2 is not None else []))
            ipart = None
        if ipart is not None and (_deferred is None or not defer_ok):
            wsum, red = colsum_pair(wpart, ipart, split_last_b=True)
        elif ipart is not None:
            # the input-affine sums feed the producer's backward now; the weight sums join the end-of-backward launch
            red = colsum(ipart, split_last=True)
            wsum = param_colsum(wpart, True)
        else:
            wsum = param_colsum(wpart, defer_ok)
        dw = wsum[:Co * Ci].view(wshape)
        db = wsum[Co * Ci:] if has_bias else None
        ds1 = dh1 = ds2 = dh2 = None
        if ipart is not None:
            if s1 is not None:
                ds1, dh1 = red[0], red[1]
            if s2 is not None:
                ds2, dh2 = red[2], red[1]
        if dgamma is not None:
            dgamma = dgamma[:n_affine] if gamma is not None else None
            dbeta = dbeta[:n_affine] if has_beta else None
        return (dx1, ds1, dh1, dx2, ds2, dh2, None, dw, db, None, None, dgamma, dbeta, None, None, None, None, None)


class _PwConvGroup(torch.autograd.Function):
    """K <= 3 1x1 convs of ONE shape in one launch each way: z_k = W_k . (x_k * s_k + h_k), no bias, second stream,
    statistics or stride — CTR-GCN's conv4 per subset in its one-conv form (ctr_topology below), each of which alone leaves
    the chip under-filled (157 position groups x 1..4 row blocks).  forward(K, slots, x_0.., s_0.., h_0.., w_0..) -> z_0..;
    backward: ONE grouped data gradient (dx_k, the input-scale rows), the weight gradients per conv, the partial-row sums
    through _PwConv._finish (same deferral rules: the `sink` decision of _CtrWPrep)."""

    @staticmethod
    def forward(ctx, K, slots, *ts):
        xs, ss, hs, ws = ts[:K], ts[K:2 * K], ts[2 * K:3 * K], ts[3 * K:4 * K]
        _require_cuda(*xs, *ws)
        xs = [_f32c(t) for t in xs]
        ss = [_f32c(t) for t in ss]
        hs = [_f32c(t) for t in hs]
        w2 = [_f32c(w.reshape(w.shape[0], -1)) for w in ws]
        n, Ci, T, V = xs[0].shape
        Co = w2[0].shape[0]
        zs = [sl.t for sl in slots]
        for z in zs:
            if tuple(z.shape) != (n, Co, T, V) or z.dtype != torch.float32 or not z.is_contiguous() or z.requires_grad:
                raise ValueError(f'pwconv_group(out=): expected contiguous fp32 {(n, Co, T, V)} slices without history')
        rc = native.lib().dsgcn_pwconv_fwd_group(_ptr_array(xs), _ptr_array(ss), _ptr_array(hs), 0, _ptr_array(w2),
                                                 _ptr_array(zs), K, n, Ci, Co, T, V, _stream())
        native.check(rc, 'dsgcn_pwconv_fwd_group')
        ctx.save_for_backward(*xs, *ss, *hs, *w2)
        ctx.K = K
        ctx.wshapes = [tuple(w.shape) for w in ws]
        ctx.defer_ok = [_leafish(w) for w in ws]
        sinks = []
        for w, s1 in zip(ws, ss):
            sink = getattr(w, '_dsgcn_sink', None)
            if sink is not None and not (s1 is None or getattr(s1, '_dsgcn_sink', None) is sink):
                sink = None
            sinks.append(sink)
        ctx.sinks = sinks
        ctx.set_materialize_grads(False)
        return tuple(zs)

    @staticmethod
    def backward(ctx, *gzs):
        K = ctx.K
        sv = ctx.saved_tensors
        xs, ss, hs, w2 = sv[:K], sv[K:2 * K], sv[2 * K:3 * K], sv[3 * K:4 * K]
        n, Ci, T, V = xs[0].shape
        Co = w2[0].shape[0]
        dev = xs[0].device
        lib = native.lib()
        st = _stream()
        gzs = [_f32c(g) if g is not None else torch.zeros((n, Co, T, V), device=dev, dtype=torch.float32) for g in gzs]
        dxs = [torch.empty_like(x) for x in xs]
        has_s = ss[0] is not None
        rows = lib.dsgcn_pwconv_ipart_rows(n, Ci, Co, T, V, 1) if has_s else 0
        iparts = [torch.empty((rows, Ci, 3), device=dev, dtype=torch.float32) if has_s else None for _ in range(K)]
        rc = lib.dsgcn_pwconv_dgrad_group(_ptr_array(xs), _ptr_array(ss), _ptr_array(hs), 0, _ptr_array(w2), _ptr_array(gzs),
                                          _ptr_array(dxs), _ptr_array(iparts), K, n, Ci, Co, T, V, st)
        native.check(rc, 'dsgcn_pwconv_dgrad_group')
        splits = lib.dsgcn_pwconv_wgrad_splits(n, Ci, Co, T, V, 1)
        pstride = Co * Ci + Co
        out_x, out_s, out_h, out_w = [], [], [], []
        wparts = [torch.empty((splits, pstride), device=dev, dtype=torch.float32) for _ in range(K)]
        dwp = (_ct.c_void_p * K)(*[w.data_ptr() for w in wparts])
        dbp = (_ct.c_void_p * K)(*[w.data_ptr() + 4 * Co * Ci for w in wparts])
        rc = -2
        if GROUP_WGRAD:
            rc = lib.dsgcn_pwconv_wgrad_group(_ptr_array(xs), _ptr_array(ss), _ptr_array(hs), 0, _ptr_array(gzs), dwp, dbp,
                                              pstride, K, n, Ci, Co, T, V, st)
        if rc == -2:                                 # not on the blocked kernels: one by one
            for k in range(K):
                rc = lib.dsgcn_pwconv_wgrad(_ptr(xs[k]), _ptr(ss[k]), _ptr(hs[k]), None, None, None, 0, None, None,
                                            _ptr(gzs[k]), None, None, None, wparts[k].data_ptr(),
                                            wparts[k].data_ptr() + 4 * Co * Ci, pstride, n, Ci, Co, T, V, 1, 0, st)
                native.check(rc, 'dsgcn_pwconv_wgrad')
        else:
            native.check(rc, 'dsgcn_pwconv_wgrad_group')
        for k in range(K):
            wpart = wparts[k]
            sink = ctx.sinks[k] is not None and bool(ctx.sinks[k])
            r = _PwConv._finish(wpart, iparts[k], dxs[k], None, ss[k], None, Co, Ci, ctx.wshapes[k], False, None, None, None,
                                False, 0, ctx.defer_ok[k], None, None, sink)
            out_x.append(r[0]); out_s.append(r[1]); out_h.append(r[2]); out_w.append(r[7])
        return (None, None, *out_x, *out_s, *out_h, *out_w)


def pwconv_group(xs, affs, ws, slots):
    """The K convs ``pwconv(x_k, aff_k, None, None, False, w_k, None, out=slot_k)`` as one launch each way when the shape
    takes the grouped form, one by one otherwise.  -> [z_k]"""
    K = len(xs)
    n, Ci, T, V = xs[0].shape
    Co = ws[0].shape[0]
    same = all(x.shape == xs[0].shape for x in xs) and all(w.shape == ws[0].shape for w in ws)
    if (GROUP_CONVS and 2 <= K <= 3 and same and xs[0].is_cuda and
            native.lib().dsgcn_pwconv_group_ok(n, Ci, Co, T, V) == 1):
        ss = [a[0] if a is not None else None for a in affs]
        hs = [a[1] if a is not None else None for a in affs]
        return list(_PwConvGroup.apply(K, list(slots), *xs, *ss, *hs, *ws))
    return [pwconv(x, a, None, None, False, w, None, 1, False, out=sl)[0] for x, a, w, sl in zip(xs, affs, ws, slots)]


GROUP_CONVS = _os.environ.get('DSGCN_GROUP_CONVS', '1') == '1'
GROUP_WGRAD = _os.environ.get('DSGCN_GROUP_WGRAD', '1') == '1'      # the group's weight gradients in one launch too


class OutSlot:
    """A preallocated output (a slice of a larger buffer) for ``pwconv(..., out=OutSlot(t))``: the kernel writes z there
    instead of a fresh tensor.  A wrapper, not a tensor argument: autograd must not see the buffer as an input."""
    __slots__ = ('t',)

    def __init__(self, t):
        self.t = t


def pwconv(x1, a1, x2, a2, relu, weight, bias, stride=1, aug=False, gamma=None, beta=None, eps=1e-5, n_affine=None,
           want_bn=False, out=None):
    """-> (z, zaug, scale, shift, mean, var); scale/shift/mean/var are None unless want_bn."""
    s1, h1 = a1 if a1 is not None else (None, None)
    s2, h2 = a2 if a2 is not None else (None, None)
    if n_affine is None:
        n_affine = weight.shape[0] if gamma is not None else 0
    bn = BNCtx() if want_bn else None
    out = _PwConv.apply(x1, s1, h1, x2, s2, h2, bool(relu), weight, bias, int(stride), bool(aug), gamma, beta,
                        float(eps), int(n_affine), bool(want_bn), bn, out)
    if want_bn:
        Tout = out[0].shape[2]
        count = float(x1.shape[0] * Tout * (x1.shape[3] + (1 if aug else 0)))
        _bn_attach(out[2], bn, out[4], out[5], gamma, eps, count, n_affine)
    return out


# ---------------------------------------------------------------------------------------------
# K-D  temporal units
# ---------------------------------------------------------------------------------------------

def tmean(x, ld=True):
    """Mean over frames of the network input (n,C,T,V) -> (n,C,V): the x.mean(-2) of the first block's adjacency
    (gcn.py:2246 / gcn.py:651); later blocks get it from the previous block's fuse_out.  ld: an int >= V pads the joint
    rows with zeros to that length (the layout ``dynadj`` reads: no pad launches in front of it)."""
    _require_cuda(x)
    return fuse_out(x, None, None, None, 0, ld)[1]


class _BranchAct(torch.autograd.Function):
    """h (n,C,T,V+1) = act(z*scale+shift) with the global-joint column appended (ReLU on channels < n_act);
    zaug=None: no extra column, h (n,C,T,V)."""

    @staticmethod
    def forward(ctx, z, zaug, scale, shift, n_act):
        _require_cuda(z)
        z, zaug, scale, shift = [_f32c(t) for t in (z, zaug, scale, shift)]
        n, C, T, V = z.shape
        h = torch.empty((n, C, T, V + (1 if zaug is not None else 0)), device=z.device, dtype=torch.float32)
        rc = native.lib().dsgcn_branch_act_fwd(_ptr(z), _ptr(zaug), _ptr(scale), _ptr(shift), int(n_act), _ptr(h), n, C,
                                               T, V, _stream())
        native.check(rc, 'dsgcn_branch_act_fwd')
        ctx.save_for_backward(z, zaug, scale, shift)
        ctx.n_act = int(n_act)
        ctx.bn = _bn_of(scale)
        return h

    @staticmethod
    def backward(ctx, dh):
        z, zaug, scale, shift = ctx.saved_tensors
        n, C, T, V = z.shape
        dh = _f32c(dh)
        dz = torch.empty_like(z)
        dzaug = torch.empty_like(zaug) if zaug is not None else None
        part = torch.empty((n, C, 2), device=z.device, dtype=torch.float32)
        rc = native.lib().dsgcn_branch_act_bwd(_ptr(z), _ptr(zaug), _ptr(scale), _ptr(shift), ctx.n_act, _ptr(dh),
                                               _ptr(dz), _ptr(dzaug), _ptr(part), n, C, T, V, _stream())
        native.check(rc, 'dsgcn_branch_act_bwd')
        if ctx.bn is not None:
            _bn_feed(ctx.bn, part, 2, 0, 1)
            return dz, dzaug, None, None, None
        red = colsum(part, split_last=True)
        return dz, dzaug, red[0], red[1], None


class _TmsCombine(torch.autograd.Function):
    """f = o[..., :V] + o[..., V] * coeff, with the train-mode BN of f as a deferred affine (like _PwConv)."""

    @staticmethod
    def forward(ctx, o, coeff, gamma, beta, eps, want_bn, bn=None):
        _require_cuda(o)
        ctx.bn = bn
        o, coeff, gamma, beta = [_f32c(t) for t in (o, coeff, gamma, beta)]
        n, C, T, V1 = o.shape
        V = V1 - 1
        dev = o.device
        f = torch.empty((n, C, T, V), device=dev, dtype=torch.float32)
        partial = torch.empty((n, C, 2), device=dev, dtype=torch.float32) if want_bn else None
        lib = native.lib()
        rc = lib.dsgcn_tms_combine_fwd(_ptr(o), _ptr(coeff), _ptr(f), _ptr(partial), n, C, T, V, _stream())
        native.check(rc, 'dsgcn_tms_combine_fwd')
        scale = shift = mean = var = None
        count = float(n * T * V)
        if want_bn:
            stats = torch.empty((4, C), device=dev, dtype=torch.float32)
            mean, var, scale, shift = stats[0], stats[1], stats[2], stats[3]
            rc = lib.dsgcn_bn_finalize(_ptr(partial), n, C, count, _ptr(gamma), _ptr(beta), float(eps), _ptr(mean),
                                       _ptr(var), _ptr(scale), _ptr(shift), C, _stream())
            native.check(rc, 'dsgcn_bn_finalize')
            ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(o, coeff, gamma, mean, var)
        ctx.cfg = (float(eps), bool(want_bn), count, beta is not None)
        ctx.defer_ok = _leafish(coeff)
        return f, scale, shift, mean, var

    @staticmethod
    def backward(ctx, gf, gscale, gshift, _gm, _gv):
        o, coeff, gamma, mean, var = ctx.saved_tensors
        eps, want_bn, count, has_beta = ctx.cfg
        n, C, T, V1 = o.shape
        V = V1 - 1
        dev = o.device
        lib = native.lib()
        gf, gscale, gshift = _f32c(gf), _f32c(gscale), _f32c(gshift)
        A0 = B0 = dgamma = dbeta = None
        if want_bn:
            dgamma, dbeta, A0, B0 = _bn_coef(ctx.bn, gscale, gshift, mean, var, gamma, eps, count, C, C)
        do = torch.empty_like(o)
        pcoef = torch.empty((n * C, V), device=dev, dtype=torch.float32)
        rc = lib.dsgcn_tms_combine_bwd(_ptr(o), _ptr(coeff), _ptr(gf), _ptr(A0), _ptr(B0), _ptr(do), _ptr(pcoef), n, C,
                                       T, V, _stream())
        native.check(rc, 'dsgcn_tms_combine_bwd')
        dcoeff = param_colsum(pcoef, ctx.defer_ok)
        if dgamma is not None:
            dgamma = dgamma if gamma is not None else None
            dbeta = dbeta if has_beta else None
        return do, dcoeff, dgamma, dbeta, None, None, None


def _int_array(vals):
    return (_ct.c_int * len(vals))(*vals)


def _ptr_array(tensors):
    return (_ct.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


class _TapBranches(torch.autograd.Function):
    """Temporal windows over h (n,Cin,T,V1) -> o (n,Cout,T',V1): (KT,1) convs on the matrix core, (3,1) max-pool,
    strided copy — one HIP launch per direction.  Window i reads channels [ci0,ci0+cin) and writes [co0,co0+cout)."""

    @staticmethod
    def forward(ctx, h, stride, KT, Cout, types, ci0s, co0s, cins, couts, dils, *wb):
        _require_cuda(h)
        h = _f32c(h)
        n, Cin, T, V1 = h.shape
        nbr = len(types)
        ws = [_f32c(t) for t in wb[:nbr]]
        bs = [_f32c(t) for t in wb[nbr:]]
        Tout = (T + stride - 1) // stride
        o = torch.empty((n, Cout, Tout, V1), device=h.device, dtype=torch.float32)
        rc = native.lib().dsgcn_tapconv_fwd(_ptr(h), _ptr(o), n, Cin, Cout, T, V1, stride, KT, nbr, _int_array(types),
                                            _int_array(ci0s), _int_array(co0s), _int_array(cins), _int_array(couts),
                                            _int_array(dils), _ptr_array(ws), _ptr_array(bs), _stream())
        native.check(rc, 'dsgcn_tapconv_fwd')
        ctx.save_for_backward(h, *[w for w in ws if w is not None])
        ctx.cfg = (stride, KT, Cout, tuple(types), tuple(ci0s), tuple(co0s), tuple(cins), tuple(couts), tuple(dils),
                   tuple(b is not None for b in bs))
        ctx.defer_ok = _leafish(*wb)
        return o

    @staticmethod
    def backward(ctx, go):
        h, *wsaved = ctx.saved_tensors
        stride, KT, Cout, types, ci0s, co0s, cins, couts, dils, has_b = ctx.cfg
        n, Cin, T, V1 = h.shape
        nbr = len(types)
        go = _f32c(go)
        lib = native.lib()
        it = iter(wsaved)
        ws = [next(it) if t == 0 else None for t in types]
        tabs = (_int_array(types), _int_array(ci0s), _int_array(co0s), _int_array(cins), _int_array(couts),
                _int_array(dils))
        dh = torch.empty_like(h)
        rc = lib.dsgcn_tapconv_dgrad(_ptr(h), _ptr(go), _ptr(dh), n, Cin, Cout, T, V1, stride, KT, nbr, *tabs,
                                     _ptr_array(ws), _stream())
        native.check(rc, 'dsgcn_tapconv_dgrad')
        Tout = go.shape[2]
        if 0 not in types:                 # pooling / pass-through windows only: nothing to learn
            return (dh, None, None, None, None, None, None, None, None, None, *([None] * (2 * nbr)))
        offs, off = [], 0
        for t, ci, co in zip(types, cins, couts):
            offs.append(off)
            if t == 0:
                off += co * ci * KT + co
        pstride = max(off, 1)
        # k-splits: enough blocks to fill the chip, but keep the partial buffer (splits x pstride floats) around 8 MB
        wmax = max([max(ci, co) for t, ci, co in zip(types, cins, couts) if t == 0] or [0])
        pref = lib.dsgcn_tapconv_wgrad_splits(n, Cin, Cout, T, V1, stride, KT, nbr, tabs[0], tabs[3], tabs[4], tabs[5])
        if pref > 0:        # wide-load kernel (csrc/tapconv.hip k_tapw): one partial row per workgroup
            splits = pref
        elif wmax <= 32:    # narrow windows: every wave is its own split (csrc/tapconv.hip k_tapconv_wgrad_narrow)
            splits = max(64, min(1024, (1 << 21) // pstride)) // 4 * 4
            splits = max(4, min(splits, (n * Tout) // 4 * 4))
        else:
            # wide windows: ~512 workgroups over (64x64 tiles) x (K-splits), partial buffer capped at 64 MB
            tiles = sum(((ci + 63) // 64) * ((co + 63) // 64) for t, ci, co in zip(types, cins, couts) if t == 0)
            splits = max(16, min(256, 512 // max(tiles, 1), (1 << 24) // pstride))
            splits = max(1, min(splits, n * ((Tout + 1) // 2)))
        part = torch.empty((splits, pstride), device=h.device, dtype=torch.float32)
        base = part.data_ptr()
        dwp = (_ct.c_void_p * nbr)(*[base + 4 * o if t == 0 else None for t, o in zip(types, offs)])
        dbp = (_ct.c_void_p * nbr)(*[base + 4 * (o + co * ci * KT) if t == 0 else None
                                     for t, o, ci, co in zip(types, offs, cins, couts)])
        rc = lib.dsgcn_tapconv_wgrad(_ptr(h), _ptr(go), n, Cin, Cout, T, V1, stride, KT, nbr, *tabs, dwp, dbp, splits,
                                     pstride, _stream())
        native.check(rc, 'dsgcn_tapconv_wgrad')
        red = param_colsum(part, ctx.defer_ok)
        dws = [red[o:o + co * ci * KT].view(co, ci, KT, 1) if t == 0 else None
               for t, o, ci, co in zip(types, offs, cins, couts)]
        dbs = [red[o + co * ci * KT:o + co * ci * KT + co] if (t == 0 and hb) else None
               for t, o, ci, co, hb in zip(types, offs, cins, couts, has_b)]
        return (dh, None, None, None, None, None, None, None, None, None, *dws, *dbs)


def _branch_tables(branch_cfg, widths, conv_w, conv_b):
    """ms_cfg-style branch list -> window tables of _TapBranches (input and output windows coincide)."""
    types, c0s, bcs, dils, ws, bs = [], [], [], [], [], []
    c0 = ci = 0
    KT = None
    for cfg, bc in zip(branch_cfg, widths):
        if cfg == '1x1':
            types.append(2); dils.append(1); ws.append(None); bs.append(None)
        elif cfg[0] == 'max':
            if cfg[1] != 3:
                raise NotImplementedError('max-pool branch: only kernel 3 has a HIP path')
            types.append(1); dils.append(1); ws.append(None); bs.append(None)
        else:
            k, d = cfg
            if k not in (3, 5, 9) or (KT is not None and k != KT):
                raise NotImplementedError('temporal conv branches: HIP path covers one kernel size of 3, 5 or 9 per unit')
            KT = k
            types.append(0); dils.append(int(d)); ws.append(conv_w[ci]); bs.append(conv_b[ci])
            ci += 1
        c0s.append(c0); bcs.append(int(bc))
        c0 += bc
    return KT or 3, types, c0s, bcs, dils, ws, bs


class _TemporalFused(torch.autograd.Function):
    """The whole multi-scale temporal stage between the unit's two 1x1 convs as ONE launch per direction (csrc/tms.hip):
    BN affine + ReLU of the branch conv's output applied while staging, the global-joint column carried in LDS only,
    dilated convs / max-pool / strided copy per channel window, ``f = o[..., :V] + o[..., V] * coeff`` and the batch
    statistics of f in the epilogue.  Neither h nor o is materialised.  zaug / coeff None: no global joint (mstcn, MSTCN)."""

    @staticmethod
    def forward(ctx, z, zaug, scale, shift, coeff, gamma, beta, n_act, stride, KT, types, c0s, bcs, dils, eps, want_bn,
                bn, *wb):
        _require_cuda(z)
        ctx.bn, ctx.bn_in = bn, _bn_of(scale)
        z, zaug, scale, shift, coeff, gamma, beta = [_f32c(t) for t in (z, zaug, scale, shift, coeff, gamma, beta)]
        nbr = len(types)
        ws = [_f32c(t) for t in wb[:nbr]]
        bs = [_f32c(t) for t in wb[nbr:]]
        n, C, T, V = z.shape
        Tout = (T + stride - 1) // stride
        dev = z.device
        lib = native.lib()
        aug = zaug is not None
        tabs = (_int_array(types), _int_array(c0s), _int_array(bcs), _int_array(dils))
        rows = lib.dsgcn_tms_rows(0, n, C, T, V, stride, KT, nbr, tabs[0], tabs[2], tabs[3], int(aug))
        assert rows > 0
        f = torch.empty((n, C, Tout, V), device=dev, dtype=torch.float32)
        oaug = torch.empty((n, C, Tout), device=dev, dtype=torch.float32) if aug else None
        stats = torch.empty((rows, C, 2), device=dev, dtype=torch.float32) if want_bn else None
        rc = lib.dsgcn_tms_fwd(_ptr(z), _ptr(zaug), _ptr(scale), _ptr(shift), int(n_act), _ptr(coeff), _ptr(f),
                               _ptr(oaug), _ptr(stats), n, C, T, V, stride, KT, nbr, *tabs, _ptr_array(ws),
                               _ptr_array(bs), _stream())
        native.check(rc, 'dsgcn_tms_fwd')
        scale1 = shift1 = mean = var = None
        count = float(n * Tout * V)
        if want_bn:
            st = torch.empty((4, C), device=dev, dtype=torch.float32)
            mean, var, scale1, shift1 = st[0], st[1], st[2], st[3]
            rc = lib.dsgcn_bn_finalize(_ptr(stats), rows, C, count, _ptr(gamma), _ptr(beta), float(eps), _ptr(mean),
                                       _ptr(var), _ptr(scale1), _ptr(shift1), C, _stream())
            native.check(rc, 'dsgcn_bn_finalize')
            ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(z, zaug, scale, shift, coeff, f, oaug, gamma, mean, var, *[w for w in ws if w is not None])
        ctx.cfg = (int(n_act), int(stride), int(KT), tuple(types), tuple(c0s), tuple(bcs), tuple(dils), float(eps),
                   bool(want_bn), count, beta is not None, tuple(b is not None for b in bs))
        ctx.defer_ok = _leafish(coeff, *wb)
        return f, scale1, shift1, mean, var

    @staticmethod
    def backward(ctx, gf, gscale, gshift, _gm, _gv):
        z, zaug, scale, shift, coeff, f, oaug, gamma, mean, var, *wsaved = ctx.saved_tensors
        n_act, stride, KT, types, c0s, bcs, dils, eps, want_bn, count, has_beta, has_b = ctx.cfg
        n, C, T, V = z.shape
        nbr = len(types)
        dev = z.device
        lib = native.lib()
        st = _stream()
        aug = zaug is not None
        gf, gscale, gshift = _f32c(gf), _f32c(gscale), _f32c(gshift)
        A0 = B0 = dgamma = dbeta = None
        if want_bn:
            dgamma, dbeta, A0, B0 = _bn_coef(ctx.bn, gscale, gshift, mean, var, gamma, eps, count, C, C)
        if gf is None:
            gf = torch.zeros_like(f)
        it = iter(wsaved)
        ws = [next(it) if t == 0 else None for t in types]
        tabs = (_int_array(types), _int_array(c0s), _int_array(bcs), _int_array(dils))
        rows = lib.dsgcn_tms_rows(1, n, C, T, V, stride, KT, nbr, tabs[0], tabs[2], tabs[3], int(aug))
        paff = torch.empty((rows, C, 2), device=dev, dtype=torch.float32)
        pcoeff = torch.empty((rows * nbr, V), device=dev, dtype=torch.float32) if aug else None
        dz = torch.empty_like(z)
        dzaug = torch.empty_like(zaug) if aug else None
        rc = lib.dsgcn_tms_dgrad(_ptr(z), _ptr(zaug), _ptr(scale), _ptr(shift), n_act, _ptr(coeff), _ptr(gf), _ptr(f),
                                 _ptr(A0), _ptr(B0), _ptr(oaug), _ptr(dz), _ptr(dzaug), _ptr(paff), _ptr(pcoeff), n, C, T,
                                 V, stride, KT, nbr, *tabs, _ptr_array(ws), st)
        native.check(rc, 'dsgcn_tms_dgrad')
        dscale = dshift = dcoeff = None
        if scale is not None and ctx.bn_in is not None:
            _bn_feed(ctx.bn_in, paff, 2, 0, 1)
        elif scale is not None:
            red = colsum(paff, split_last=True)
            dscale, dshift = red[0], red[1]
        if aug:
            dcoeff = param_colsum(pcoeff, ctx.defer_ok)
        dws, dbs = [None] * nbr, [None] * nbr
        if 0 in types:
            offs, off = [], 0
            for t, bc in zip(types, bcs):
                offs.append(off)
                if t == 0:
                    off += bc * bc * KT + bc
            pstride = off
            rows2 = lib.dsgcn_tms_rows(2, n, C, T, V, stride, KT, nbr, tabs[0], tabs[2], tabs[3], int(aug))
            part = torch.empty((rows2, pstride), device=dev, dtype=torch.float32)
            base = part.data_ptr()
            dwp = (_ct.c_void_p * nbr)(*[base + 4 * o if t == 0 else None for t, o in zip(types, offs)])
            dbp = (_ct.c_void_p * nbr)(*[base + 4 * (o + bc * bc * KT) if t == 0 else None
                                         for t, o, bc in zip(types, offs, bcs)])
            rc = lib.dsgcn_tms_wgrad(_ptr(z), _ptr(zaug), _ptr(scale), _ptr(shift), n_act, _ptr(coeff), _ptr(gf), _ptr(f),
                                     _ptr(A0), _ptr(B0), n, C, T, V, stride, KT, nbr, *tabs, dwp, dbp, pstride, st)
            native.check(rc, 'dsgcn_tms_wgrad')
            red = param_colsum(part, ctx.defer_ok)
            dws = [red[o:o + bc * bc * KT].view(bc, bc, KT, 1) if t == 0 else None for t, o, bc in zip(types, offs, bcs)]
            dbs = [red[o + bc * bc * KT:o + bc * bc * KT + bc] if (t == 0 and hb) else None
                   for t, o, bc, hb in zip(types, offs, bcs, has_b)]
        if dgamma is not None:
            dgamma = dgamma if gamma is not None else None
            dbeta = dbeta if has_beta else None
        return (dz, dzaug, dscale, dshift, dcoeff, dgamma, dbeta, None, None, None, None, None, None, None, None, None,
                None, *dws, *dbs)


class _TemporalSplit(torch.autograd.Function):
    """dgmstcn's temporal stage on the split layout (csrc/tmsplit.hip): the V joint columns through the matrix-core window
    kernels straight from z to f (BatchNorm affine + ReLU and the global-joint term while loading, the statistics of f in
    the epilogue), the global-joint column as extra blocks of the same launches on (n, C, T) tensors.  The (V+1)-column
    tensors of tcn.py:409-420 (h, o and their gradients) and the branch_act / combine passes do not exist; backward
    materialises ``ge = gf + A0 + B0*f`` once (it feeds the data AND the weight gradient) and finishes ``dz`` / the branch
    BatchNorm's sums in the data gradient's epilogue.  stride 1 or 2 over frames."""
    FRONT = 32           # floats in front of ge / doaug: the stride-2 data gradient may start a run before a plane

    @staticmethod
    def forward(ctx, z, zaug, scale, shift, coeff, gamma, beta, n_act, stride, types, c0s, bcs, dils, eps, want_bn, bn, *wb):
        _require_cuda(z)
        ctx.bn = bn
        z, zaug, scale, shift, coeff, gamma, beta = [_f32c(t) for t in (z, zaug, scale, shift, coeff, gamma, beta)]
        nbr = len(types)
        ws = [_f32c(t) for t in wb[:nbr]]
        bs = [_f32c(t) for t in wb[nbr:]]
        n, C, T, V = z.shape
        To = T // stride
        dev = z.device
        lib = native.lib()
        tabs = (_int_array(types), _int_array(c0s), _int_array(bcs), _int_array(dils))
        rows = lib.dsgcn_tms_split_rows(0, n, C, T, V, stride, 3, nbr, *tabs)
        assert rows > 0
        f = torch.empty((n, C, To, V), device=dev, dtype=torch.float32)
        oaug = torch.empty((n, C, To), device=dev, dtype=torch.float32)
        stats = torch.empty((rows, C, 2), device=dev, dtype=torch.float32) if want_bn else None
        rc = lib.dsgcn_tms_split_fwd(_ptr(z), _ptr(zaug), _ptr(scale), _ptr(shift), int(n_act), _ptr(coeff), _ptr(f),
                                     _ptr(oaug), _ptr(stats), n, C, T, V, stride, nbr, *tabs, _ptr_array(ws), _ptr_array(bs),
                                     _stream())
        native.check(rc, 'dsgcn_tms_split_fwd')
        scale1 = shift1 = mean = var = None
        count = float(n * To * V)
        if want_bn:
            st = torch.empty((4, C), device=dev, dtype=torch.float32)
            mean, var, scale1, shift1 = st[0], st[1], st[2], st[3]
            rc = lib.dsgcn_bn_finalize(_ptr(stats), rows, C, count, _ptr(gamma), _ptr(beta), float(eps), _ptr(mean),
                                       _ptr(var), _ptr(scale1), _ptr(shift1), C, _stream())
            native.check(rc, 'dsgcn_bn_finalize')
            ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(z, zaug, scale, shift, coeff, f, oaug, gamma, mean, var, *[w for w in ws if w is not None])
        ctx.cfg = (int(n_act), int(stride), tuple(types), tuple(c0s), tuple(bcs), tuple(dils), float(eps), bool(want_bn),
                   count, beta is not None, tuple(b is not None for b in bs))
        ctx.bn_in = _bn_of(scale)
        ctx.defer_ok = _leafish(coeff, *wb)
        return f, scale1, shift1, mean, var

    @staticmethod
    def backward(ctx, gf, gscale, gshift, _gm, _gv):
        z, zaug, scale, shift, coeff, f, oaug, gamma, mean, var, *wsaved = ctx.saved_tensors
        n_act, stride, types, c0s, bcs, dils, eps, want_bn, count, has_beta, has_b = ctx.cfg
        n, C, T, V = z.shape
        To = f.shape[2]
        nbr = len(types)
        dev = z.device
        lib = native.lib()
        st = _stream()
        gf, gscale, gshift = _f32c(gf), _f32c(gscale), _f32c(gshift)
        A0 = B0 = dgamma = dbeta = None
        if want_bn:
            dgamma, dbeta, A0, B0 = _bn_coef(ctx.bn, gscale, gshift, mean, var, gamma, eps, count, C, C)
        FR = _TemporalSplit.FRONT
        gbuf = torch.empty(FR + f.numel() + FR + n * C * To, device=dev, dtype=torch.float32)
        ge = gbuf[FR:FR + f.numel()].view(f.shape)
        doaug = gbuf[2 * FR + f.numel():].view(n, C, To)
        pcoef = torch.empty((n * C, V), device=dev, dtype=torch.float32)
        rc = lib.dsgcn_tms_split_prep(_ptr(gf), _ptr(f), _ptr(oaug), _ptr(coeff), _ptr(A0), _ptr(B0), _ptr(ge), _ptr(doaug),
                                      _ptr(pcoef), n, C, To, V, st)
        native.check(rc, 'dsgcn_tms_split_prep')
        dcoeff = param_colsum(pcoef, ctx.defer_ok)
        it = iter(wsaved)
        ws = [next(it) if t == 0 else None for t in types]
        tabs = (_int_array(types), _int_array(c0s), _int_array(bcs), _int_array(dils))
        rows = lib.dsgcn_tms_split_rows(1, n, C, T, V, stride, 3, nbr, *tabs)
        part = torch.empty((rows, C, 2), device=dev, dtype=torch.float32)
        dz = torch.empty_like(z)
        dzaug = torch.empty_like(zaug)
        rc = lib.dsgcn_tms_split_dgrad(_ptr(z), _ptr(zaug), _ptr(scale), _ptr(shift), n_act, _ptr(ge), _ptr(doaug), _ptr(dz),
                                       _ptr(dzaug), _ptr(part), n, C, T, V, stride, nbr, *tabs, _ptr_array(ws), st)
        native.check(rc, 'dsgcn_tms_split_dgrad')
        dscale = dshift = None
        hosted = []
        if ctx.bn_in is not None:
            # the branch BatchNorm's coefficient job rides in the weight gradient's launch below (nothing waits for that one)
            hosted = _bn_feed_multi([(ctx.bn_in, part, 2, 0, 1)], launch=False)
        elif scale is not None:
            red = colsum(part, split_last=True)
            dscale, dshift = red[0], red[1]
        offs, off = [], 0
        for t, bc in zip(types, bcs):
            offs.append(off)
            if t == 0:
                off += bc * bc * 3 + bc
        pstride = off
        splits = lib.dsgcn_tms_split_rows(2, n, C, T, V, stride, 3, nbr, *tabs)
        wpart = torch.empty((splits, pstride), device=dev, dtype=torch.float32)
        base = wpart.data_ptr()
        dwp = (_ct.c_void_p * nbr)(*[base + 4 * o if t == 0 else None for t, o in zip(types, offs)])
        dbp = (_ct.c_void_p * nbr)(*[base + 4 * (o + bc * bc * 3) if t == 0 else None
                                     for t, o, bc in zip(types, offs, bcs)])
        rc = lib.dsgcn_tms_split_wgrad_jobs(_ptr(z), _ptr(zaug), _ptr(scale), _ptr(shift), n_act, _ptr(ge), _ptr(doaug), n, C,
                                            T, V, stride, nbr, *tabs, dwp, dbp, splits, pstride, *_job_array(hosted), st)
        native.check(rc, 'dsgcn_tms_split_wgrad_jobs')
        red = param_colsum(wpart, ctx.defer_ok)
        dws = [red[o:o + bc * bc * 3].view(bc, bc, 3, 1) if t == 0 else None for t, o, bc in zip(types, offs, bcs)]
        dbs = [red[o + bc * bc * 3:o + bc * bc * 3 + bc] if (t == 0 and hb) else None
               for t, o, bc, hb in zip(types, offs, bcs, has_b)]
        if dgamma is not None:
            dgamma = dgamma if gamma is not None else None
            dbeta = dbeta if has_beta else None
        return (dz, dzaug, dscale, dshift, dcoeff, dgamma, dbeta, None, None, None, None, None, None, None, None, None,
                *dws, *dbs)


def _split_temporal(z, zaug, scale, shift, coeff, n_act, branch_cfg, widths, conv_w, conv_b, stride, gamma, beta, eps,
                    want_bn):
    """-> (f, scale, shift, mean, var) through csrc/tmsplit.hip, or None when the shape is not eligible."""
    if zaug is None or any(not isinstance(c, str) and c[0] != 'max' and c[0] != 3 for c in branch_cfg):
        return None
    if int(stride) != 1 and SPLIT_TEMPORAL == '1':
        return None
    KT, types, c0s, bcs, dils, ws, bs = _branch_tables(branch_cfg, widths, conv_w, conv_b)
    n, C, T, V = z.shape
    if native.lib().dsgcn_tms_split_rows(-1, n, C, T, V, int(stride), KT, len(types), _int_array(types), _int_array(c0s),
                                         _int_array(bcs), _int_array(dils)) != 1:
        return None
    bn = BNCtx() if want_bn else None
    out = _TemporalSplit.apply(z, zaug, scale, shift, coeff, gamma, beta, int(n_act), int(stride), types, c0s, bcs, dils,
                               float(eps), bool(want_bn), bn, *ws, *bs)
    if want_bn:
        _bn_attach(out[1], bn, out[3], out[4], gamma, eps, float(n * (T // int(stride)) * V), C)
    return out


def _fused_temporal(z, zaug, scale, shift, coeff, n_act, branch_cfg, widths, conv_w, conv_b, stride, gamma, beta, eps,
                    want_bn):
    """-> (f, scale, shift, mean, var) through csrc/tms.hip, or None when the shape is not eligible."""
    KT, types, c0s, bcs, dils, ws, bs = _branch_tables(branch_cfg, widths, conv_w, conv_b)
    n, C, T, V = z.shape
    if FUSED_TEMPORAL == '0' or (FUSED_TEMPORAL == 'auto' and KT == 3):
        return None
    if sum(bcs) != C or any(c0s[i] + bcs[i] != c0s[i + 1] for i in range(len(bcs) - 1)) or c0s[0] != 0:
        return None
    rows = native.lib().dsgcn_tms_rows(0, n, C, T, V, int(stride), KT, len(types), _int_array(types), _int_array(bcs),
                                       _int_array(dils), int(zaug is not None))
    if rows <= 0:
        return None
    bn = BNCtx() if want_bn else None
    out = _TemporalFused.apply(z, zaug, scale, shift, coeff, gamma, beta, int(n_act), int(stride), KT, types, c0s, bcs,
                               dils, float(eps), bool(want_bn), bn, *ws, *bs)
    if want_bn:
        Tout = (T + int(stride) - 1) // int(stride)
        _bn_attach(out[1], bn, out[3], out[4], gamma, eps, float(n * Tout * V), C)
    return out


def temporal_ms(z, zaug, scale, shift, n_act, branch_cfg, widths, conv_w, conv_b, add_coeff, stride, gamma=None,
                beta=None, eps=1e-5, want_bn=False):
    """-> (f, scale, shift, mean, var): branch_act -> temporal branches (dilated convs / max-pool / copy) -> combine
    (+ statistics of transform.0's BatchNorm); three HIP stages, no torch.cat, no MIOpen."""
    _require_cuda(z)
    n, C, T, V = z.shape
    coeff = add_coeff if add_coeff.shape[0] == V else add_coeff[:V].contiguous()
    if FUSED_TEMPORAL != '0':
        out = _fused_temporal(z, zaug, scale, shift, coeff, n_act, branch_cfg, widths, conv_w, conv_b, stride, gamma, beta,
                              eps, want_bn)
        if out is not None:
            return out
    if SPLIT_TEMPORAL != '0':
        out = _split_temporal(z, zaug, scale, shift, coeff, n_act, branch_cfg, widths, conv_w, conv_b, stride, gamma, beta,
                              eps, want_bn)
        if out is not None:
            return out
    # shapes the fused kernels do not take (plane sizes that are not multiples of 4 floats, ...): the staged form
    h = _BranchAct.apply(z, zaug, scale, shift, n_act)
    KT, types, c0s, bcs, dils, ws, bs = _branch_tables(branch_cfg, widths, conv_w, conv_b)
    o = _TapBranches.apply(h, int(stride), KT, C, types, c0s, c0s, bcs, bcs, dils, *ws, *bs)
    bn = BNCtx() if want_bn else None
    out = _TmsCombine.apply(o, coeff, gamma, beta, float(eps), bool(want_bn), bn)
    if want_bn:
        _bn_attach(out[1], bn, out[3], out[4], gamma, eps, float(o.shape[0] * o.shape[2] * (o.shape[3] - 1)), o.shape[1])
    return out


class _PlaneStats(torch.autograd.Function):
    """Train-mode BatchNorm of an already materialised tensor as a deferred affine: (scale, shift, mean, var) from the
    per-plane sums of o.  Backward adds the statistics terms A0[c] + B0[c]*o to the gradient of o."""

    @staticmethod
    def forward(ctx, o, gamma, beta, eps):
        _require_cuda(o)
        o, gamma, beta = _f32c(o), _f32c(gamma), _f32c(beta)
        n, C, T, V = o.shape
        dev = o.device
        lib = native.lib()
        partial = torch.empty((n, C, 2), device=dev, dtype=torch.float32)
        native.check(lib.dsgcn_plane_stats(_ptr(o), _ptr(partial), n * C, T * V, _stream()), 'dsgcn_plane_stats')
        stats = torch.empty((4, C), device=dev, dtype=torch.float32)
        mean, var, scale, shift = stats[0], stats[1], stats[2], stats[3]
        count = float(n * T * V)
        rc = lib.dsgcn_bn_finalize(_ptr(partial), n, C, count, _ptr(gamma), _ptr(beta), float(eps), _ptr(mean),
                                   _ptr(var), _ptr(scale), _ptr(shift), C, _stream())
        native.check(rc, 'dsgcn_bn_finalize')
        ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(o, gamma, mean, var)
        ctx.cfg = (float(eps), count, beta is not None)
        return scale, shift, mean, var

    @staticmethod
    def backward(ctx, gscale, gshift, _gm, _gv):
        o, gamma, mean, var = ctx.saved_tensors
        eps, count, has_beta = ctx.cfg
        n, C, T, V = o.shape
        lib = native.lib()
        gscale, gshift = _f32c(gscale), _f32c(gshift)
        if gscale is None and gshift is None:
            return None, None, None, None
        coef = torch.empty((4, C), device=o.device, dtype=torch.float32)
        dgamma, dbeta, A0, B0 = coef[0], coef[1], coef[2], coef[3]
        rc = lib.dsgcn_bn_bwd_coef(_ptr(gscale), _ptr(gshift), _ptr(mean), _ptr(var), _ptr(gamma), eps, count, C, C,
                                   _ptr(dgamma), _ptr(dbeta), _ptr(A0), _ptr(B0), _stream())
        native.check(rc, 'dsgcn_bn_bwd_coef')
        do = torch.empty_like(o)
        rc = lib.dsgcn_fuse_out_fwd(_ptr(o), _ptr(B0), _ptr(A0), None, None, None, 0, _ptr(do), None, n, C, T, V, V,
                                    _stream())
        native.check(rc, 'dsgcn_fuse_out_fwd')
        return do, (dgamma if gamma is not None else None), (dbeta if has_beta else None), None


def _plane_bn(o, gamma, beta, eps, want_bn):
    if not want_bn:
        return o, None, None, None, None
    return (o, *_PlaneStats.apply(o, gamma, beta, float(eps)))


def temporal_branches_bn(z, scale, shift, n_act, branch_cfg, widths, conv_w, conv_b, stride, gamma=None, beta=None,
                         eps=1e-5, want_bn=False):
    """MSTCN's temporal stage (msg3d_utils.py:84-117) after the fused branch 1x1 conv: BN+ReLU on channels < n_act,
    (k,1) dilated convs / (3,1) max-pool / strided copy per branch -> o (n,C,T',V) raw, plus the train-mode BN of o
    (the per-branch BatchNorms that close every branch, concatenated) as a deferred affine.
    -> (o, scale, shift, mean, var)"""
    _require_cuda(z)
    n, C, T, V = z.shape
    if FUSED_TEMPORAL != '0':
        out = _fused_temporal(z, None, scale, shift, None, n_act, branch_cfg, widths, conv_w, conv_b, stride, gamma, beta,
                              eps, want_bn)
        if out is not None:
            return out
    h = _BranchAct.apply(z, None, scale, shift, n_act)
    KT, types, c0s, bcs, dils, ws, bs = _branch_tables(branch_cfg, widths, conv_w, conv_b)
    o = _TapBranches.apply(h, int(stride), KT, C, types, c0s, c0s, bcs, bcs, dils, *ws, *bs)
    return _plane_bn(o, gamma, beta, eps, want_bn)


class _DwCausal(torch.autograd.Function):
    """Depthwise causal temporal taps (unitmlp's grouped Conv1d) over the mlp windows of h; zero elsewhere."""

    @staticmethod
    def forward(ctx, h, w, b, dil, stride):
        _require_cuda(h, w)
        h, w, b = _f32c(h), _f32c(w), _f32c(b)
        n, C, T, V = h.shape
        KM = w.shape[1]
        Tout = (T + stride - 1) // stride
        y = torch.empty((n, C, Tout, V), device=h.device, dtype=torch.float32)
        rc = native.lib().dsgcn_dwcausal_fwd(_ptr(h), _ptr(w), _ptr(b), _ptr(dil), _ptr(y), n, C, T, V, stride, KM,
                                             _stream())
        native.check(rc, 'dsgcn_dwcausal_fwd')
        ctx.save_for_backward(h, w, dil)
        ctx.cfg = (stride, b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, w, dil = ctx.saved_tensors
        stride, has_b = ctx.cfg
        n, C, T, V = h.shape
        KM = w.shape[1]
        dy = _f32c(dy)
        dh = torch.empty_like(h)
        part = torch.empty((n, C, 6), device=h.device, dtype=torch.float32)
        rc = native.lib().dsgcn_dwcausal_bwd(_ptr(h), _ptr(w), _ptr(dil), _ptr(dy), _ptr(dh), _ptr(part), n, C, T, V,
                                             stride, KM, _stream())
        native.check(rc, 'dsgcn_dwcausal_bwd')
        red = colsum(part)                                    # (C, 6): [dw_0 .. dw_4, db]
        return dh, red[:, :KM].contiguous(), (red[:, 5].contiguous() if has_b else None), None, None


def temporal_mlp_bn(z, scale, shift, n_act, branch_cfg, widths, conv_w, conv_b, dw_w, dw_b, dw_dil, pw_w, pw_b,
                    merge_after, stride, gamma=None, beta=None, eps=1e-5, want_bn=False):
    """msmlp's temporal stage (tcn.py:182-261 with unitmlp 525-614) after the fused branch 1x1 conv: BN+ReLU on channels
    < n_act, then per mlp window  conv1(dw(h)) + alpha * tconv(h)  (merge_after) or  conv1(dw(h) + alpha * tconv(h)),
    beside the (3,1) max-pool and the strided pass-through windows; plus the train-mode BN of the result (transform.0) as a
    deferred affine.  conv_w / conv_b: the (k,1) dilated convs ALREADY scaled by their window's alpha;
    dw_w (C, KM), dw_b (C), dw_dil (C) int32: depthwise taps per channel (dil 0 outside the mlp windows); pw_w (C, C),
    pw_b (C): the windows' conv1 as one block-diagonal channel mix (identity / zero blocks for the other windows according
    to merge_after).  -> (o, scale, shift, mean, var)"""
    _require_cuda(z)
    n, C, T, V = z.shape
    h = _BranchAct.apply(z, None, scale, shift, n_act)
    KT, types, c0s, bcs, dils, ws, bs = _branch_tables(branch_cfg, widths, conv_w, conv_b)
    o = _TapBranches.apply(h, int(stride), KT, C, types, c0s, c0s, bcs, bcs, dils, *ws, *bs)
    dw = _DwCausal.apply(h, dw_w, dw_b, dw_dil, int(stride))
    if merge_after:
        u = pwconv(dw, None, None, None, False, pw_w, pw_b, 1, False)[0]
        out = fuse_out(u, None, o, None, 0, False)[0]
    else:
        u = fuse_out(dw, None, o, None, 0, False)[0]
        out = pwconv(u, None, None, None, False, pw_w, pw_b, 1, False)[0]
    return _plane_bn(out, gamma, beta, eps, want_bn)


def temporal_unitmlp_bn(h, dw_w, dw_b, dw_dil, tw, tb, tdil, pw_w, pw_b, merge_after, stride, gamma=None, beta=None,
                        eps=1e-5, want_bn=False):
    """``unitmlp`` as a whole temporal unit (ST-GCN with tcn_type='unitmlp': stgcn.py:51-52 over tcn.py:525-614) on a
    materialised input h (n,C,T,V): depthwise causal taps (dsgcn_dwcausal), the dense (k,1) conv already scaled by alpha
    (tw / tb, None without add_tcn), the 1x1 conv after (merge_after) or before the add, plus the train-mode BN of the
    result as a deferred affine.  -> (out, scale, shift, mean, var)"""
    _require_cuda(h)
    dw = _DwCausal.apply(h, dw_w, dw_b, dw_dil, int(stride))
    t = tconv(h, tw, tb, stride, tdil)[0] if tw is not None else None
    if t is not None and merge_after:
        u = pwconv(dw, None, None, None, False, pw_w, pw_b, 1, False)[0]
        return _plane_bn(fuse_out(u, None, t, None, 0, False)[0], gamma, beta, eps, want_bn)
    u = dw if t is None else fuse_out(dw, None, t, None, 0, False)[0]
    z, _, sc, sh, mean, var = pwconv(u, None, None, None, False, pw_w, pw_b, 1, False, gamma, beta, eps, None, want_bn)
    return z, sc, sh, mean, var


def strided_frames(x, stride):
    """x (n,C,T,V) -> x[:, :, ::stride] as a contiguous tensor (tapconv's pass-through window: one HIP launch; the
    backward scatters into a zero-filled tensor)."""
    _require_cuda(x)
    C = x.shape[1]
    return _TapBranches.apply(x, int(stride), 3, C, [2], [0], [0], [C], [C], [1], None, None)


# the per-tap weight images of the dense temporal convs: one launch per step for all of them (ST-GCN: 10 -> 1)
TSPLIT_BATCH = _os.environ.get('DSGCN_TSPLIT_BATCH', '1') == '1'


def _tsplit_launch(todo):
    tab = (native.TsplitJob * len(todo))()
    for rec, job in zip(tab, todo):
        rec.w, rec.ws = job['src'][0].data_ptr(), job['img'].data_ptr()
        rec.Ci, rec.Co, rec.KT = job['dims']
    native.check(native.lib().dsgcn_tconv_wsplit_multi(tab, len(todo), _stream()), 'dsgcn_tconv_wsplit_multi')


_tconv_images = _StepBuilt(_tsplit_launch, lambda: TSPLIT_BATCH)


class _TConvGemm(torch.autograd.Function):
    """Dense (KT,1) temporal conv of a virtual input, stride 1, as a GEMM on bf16 terms (csrc/tcg.hip), with the
    statistics of the BatchNorm behind it: z = W * relu?(x1*s1+h1 (+ x2*s2+h2)) + b -> (z, scale, shift, mean, var)."""

    @staticmethod
    def forward(ctx, x1, s1, h1, x2, s2, h2, relu, weight, bias, gamma, beta, eps, want_bn, stride, bn=None):
        _require_cuda(x1, weight)
        ctx.bn, ctx.bn1, ctx.bn2 = bn, _bn_of(s1), _bn_of(s2)
        x1, s1, h1, x2, s2, h2, bias, gamma, beta = [_f32c(t) for t in (x1, s1, h1, x2, s2, h2, bias, gamma, beta)]
        w = _f32c(weight)
        n, Ci, T, V = x1.shape
        Co, _, KT, _ = w.shape
        To = (T + stride - 1) // stride
        dev = x1.device
        lib = native.lib()
        wsb = lib.dsgcn_tconv_ws_bytes(n, Ci, Co, T, V, KT, stride)
        assert wsb > 0
        ws = _tconv_images.get((Ci, Co, KT, wsb), [w], lambda: dict(
            dims=(Ci, Co, KT), img=torch.empty(wsb, device=dev, dtype=torch.uint8)))['img']
        rows = lib.dsgcn_tconv_rows(0, n, Ci, Co, T, V, KT, stride)
        z = torch.empty((n, Co, To, V), device=dev, dtype=torch.float32)
        partial = torch.empty((rows, Co, 2), device=dev, dtype=torch.float32) if want_bn else None
        rc = lib.dsgcn_tconv_fwd(_ptr(x1), _ptr(s1), _ptr(h1), _ptr(x2), _ptr(s2), _ptr(h2), int(relu), _ptr(ws),
                                 _ptr(bias), _ptr(z), _ptr(partial), n, Ci, Co, T, V, KT, stride, _stream())
        native.check(rc, 'dsgcn_tconv_fwd')
        scale = shift = mean = var = None
        count = float(n * To * V)
        if want_bn:
            stats = torch.empty((4, Co), device=dev, dtype=torch.float32)
            mean, var, scale, shift = stats[0], stats[1], stats[2], stats[3]
            rc = lib.dsgcn_bn_finalize(_ptr(partial), rows, Co, count, _ptr(gamma), _ptr(beta), float(eps), _ptr(mean),
                                       _ptr(var), _ptr(scale), _ptr(shift), Co, _stream())
            native.check(rc, 'dsgcn_bn_finalize')
            ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x1, s1, h1, x2, s2, h2, z, gamma, mean, var, ws)
        ctx.cfg = (int(relu), float(eps), bool(want_bn), count, tuple(w.shape), bias is not None, beta is not None,
                   int(stride))
        ctx.defer_ok = _leafish(weight, bias)
        return z, scale, shift, mean, var

    @staticmethod
    def backward(ctx, gz, gscale, gshift, _gm, _gv):
        x1, s1, h1, x2, s2, h2, z, gamma, mean, var, ws = ctx.saved_tensors
        relu, eps, want_bn, count, wshape, has_bias, has_beta, stride = ctx.cfg
        n, Ci, T, V = x1.shape
        Co, _, KT, _ = wshape
        dev = x1.device
        lib = native.lib()
        st = _stream()
        gz, gscale, gshift = _f32c(gz), _f32c(gscale), _f32c(gshift)
        A0 = B0 = dgamma = dbeta = None
        if want_bn:
            dgamma, dbeta, A0, B0 = _bn_coef(ctx.bn, gscale, gshift, mean, var, gamma, eps, count, Co, Co)
        if gz is None:
            gz = torch.zeros_like(z)
        dx1 = torch.empty_like(x1)
        dx2 = torch.empty_like(x2) if x2 is not None else None
        rows = lib.dsgcn_tconv_rows(1, n, Ci, Co, T, V, KT, stride)
        ipart = (torch.empty((rows, Ci, 3), device=dev, dtype=torch.float32)
                 if (s1 is not None or s2 is not None) else None)
        rc = lib.dsgcn_tconv_dgrad(_ptr(x1), _ptr(s1), _ptr(h1), _ptr(x2), _ptr(s2), _ptr(h2), relu, _ptr(ws), _ptr(z),
                                   _ptr(gz), _ptr(A0), _ptr(B0), _ptr(dx1), _ptr(dx2), _ptr(ipart), n, Ci, Co, T, V, KT,
                                   stride, st)
        native.check(rc, 'dsgcn_tconv_dgrad')
        splits = lib.dsgcn_tconv_wgrad_splits(n, Ci, Co, T, V, KT, stride)
        pstride = Co * Ci * KT + Co
        part = torch.empty((splits, pstride), device=dev, dtype=torch.float32)
        rc = lib.dsgcn_tconv_wgrad(_ptr(x1), _ptr(s1), _ptr(h1), _ptr(x2), _ptr(s2), _ptr(h2), relu, _ptr(z), _ptr(gz),
                                   _ptr(A0), _ptr(B0), part.data_ptr(), part.data_ptr() + 4 * Co * Ci * KT, pstride, n,
                                   Ci, Co, T, V, KT, stride, st)
        native.check(rc, 'dsgcn_tconv_wgrad')
        red = param_colsum(part, ctx.defer_ok)
        dw = red[:Co * Ci * KT].view(wshape)
        db = red[Co * Ci * KT:] if has_bias else None
        ds1 = dh1 = ds2 = dh2 = None
        if ipart is not None and (s1 is None or ctx.bn1 is not None) and (s2 is None or ctx.bn2 is not None):
            _bn_feed_multi(([(ctx.bn1, ipart, 3, 0, 1)] if s1 is not None else []) +
                           ([(ctx.bn2, ipart, 3, 2, 1)] if s2 is not None else []))
        elif ipart is not None:
            isum = colsum(ipart, split_last=True)
            if s1 is not None:
                ds1, dh1 = isum[0], isum[1]
            if s2 is not None:
                ds2, dh2 = isum[2], isum[1]
        if dgamma is not None:
            dgamma = dgamma if gamma is not None else None
            dbeta = dbeta if has_beta else None
        return dx1, ds1, dh1, dx2, ds2, dh2, None, dw, db, dgamma, dbeta, None, None, None, None


def tconv_gemm_ok(n, Ci, Co, T, V, KT, stride=1):
    """Does csrc/tcg.hip take this dense temporal conv shape?"""
    return bool(native.lib().dsgcn_tconv_ws_bytes(int(n), int(Ci), int(Co), int(T), int(V), int(KT), int(stride)))


def tconv_bn(x1, a1, x2, a2, relu, weight, bias, gamma=None, beta=None, eps=1e-5, want_bn=False, stride=1):
    """Dense (KT,1) temporal conv, stride 1 or 2, dilation 1, of the virtual input relu?(x1*a1 (+ x2*a2)) — no materialised
    operand, the BatchNorm statistics in the conv's epilogue.  -> (z, scale, shift, mean, var), or None when the GEMM form
    does not take the shape (the caller then materialises the input and uses `tconv`)."""
    _require_cuda(x1, weight)
    n, Ci, T, V = x1.shape
    Co, _, KT, _ = weight.shape
    if not native.lib().dsgcn_tconv_ws_bytes(n, Ci, Co, T, V, KT, int(stride)):
        return None
    s1, h1 = a1 if a1 is not None else (None, None)
    s2, h2 = a2 if a2 is not None else (None, None)
    bn = BNCtx() if want_bn else None
    out = _TConvGemm.apply(x1, s1, h1, x2, s2, h2, bool(relu), weight, bias, gamma, beta, float(eps), bool(want_bn),
                           int(stride), bn)
    if want_bn:
        _bn_attach(out[1], bn, out[3], out[4], gamma, eps, float(n * out[0].shape[2] * V), Co)
    return out


def tconv(h, weight, bias, stride, dilation, gamma=None, beta=None, eps=1e-5, want_bn=False):
    """Dense (k,1) temporal conv of a materialised tensor (unit_tcn, tcn.py:21-27) + the train-mode BN of its output
    as a deferred affine.  -> (z, scale, shift, mean, var)"""
    _require_cuda(h)
    Co, Ci, KT, _ = weight.shape
    if KT not in (3, 5, 9):
        raise NotImplementedError('dense temporal conv: HIP path covers kernel sizes 3, 5 and 9')
    z = _TapBranches.apply(h, int(stride), KT, Co, [0], [0], [0], [Ci], [Co], [int(dilation)], weight, bias)
    return _plane_bn(z, gamma, beta, eps, want_bn)


# ---------------------------------------------------------------------------------------------
# K-A'  subset-summed aggregate (ST-GCN shared A / CTR-GCN per-channel topology)
# ---------------------------------------------------------------------------------------------

class _AggSum(torch.autograd.Function):

    @staticmethod
    def forward(ctx, p, adj, K, gamma, beta, eps, want_bn, per_sample=False, bn=None):
        _require_cuda(p, adj)
        ctx.bn = bn
        p, adj, gamma, beta = _f32c(p), _f32c(adj), _f32c(gamma), _f32c(beta)
        n, KC, T, V = p.shape
        Co = KC // K
        shared = adj.dim() == 3
        if per_sample:          # (n, K, V, V): one topology per sample and subset, shared by the channels (AAGCN)
            assert adj.shape == (n, K, V, V), (adj.shape, p.shape)
            astr = (K * V * V, V * V, 0)
        elif adj.dim() == 5:    # (K, n, Co, V, V): per sample and channel, subset-major (ctr_topology's one-conv form)
            assert adj.shape == (K, n, Co, V, V), (adj.shape, p.shape)
            astr = (Co * V * V, n * Co * V * V, V * V)
        else:
            assert adj.shape == ((K, V, V) if shared else (n, KC, V, V)), (adj.shape, p.shape)
            astr = (0, V * V, 0) if shared else (KC * V * V, Co * V * V, V * V)
        dev = p.device
        lib = native.lib()
        y = torch.empty((n, Co, T, V), device=dev, dtype=torch.float32)
        prow = lib.dsgcn_aggsum_partial_rows(n, T, V)
        partial = torch.empty((prow, Co, 2), device=dev, dtype=torch.float32) if want_bn else None
        rc = lib.dsgcn_aggsum_fwd(_ptr(p), _ptr(adj), *astr, _ptr(y), _ptr(partial), n, K, Co, T, V, _stream())
        native.check(rc, 'dsgcn_aggsum_fwd')
        scale = shift = mean = var = None
        count = float(n * T * V)
        if want_bn:
            stats = torch.empty((4, Co), device=dev, dtype=torch.float32)
            mean, var, scale, shift = stats[0], stats[1], stats[2], stats[3]
            rc = lib.dsgcn_bn_finalize(_ptr(partial), prow, Co, count, _ptr(gamma), _ptr(beta), float(eps), _ptr(mean),
                                       _ptr(var), _ptr(scale), _ptr(shift), Co, _stream())
            native.check(rc, 'dsgcn_bn_finalize')
            ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(p, adj, y, gamma, mean, var)
        ctx.cfg = (K, float(eps), bool(want_bn), count, beta is not None, shared, astr, bool(per_sample))
        ctx.defer_ok = _leafish(adj) if shared else False
        return y, scale, shift, mean, var

    @staticmethod
    def backward(ctx, gy, gscale, gshift, _gm, _gv):
        p, adj, y, gamma, mean, var = ctx.saved_tensors
        K, eps, want_bn, count, has_beta, shared, astr, per_sample = ctx.cfg
        n, KC, T, V = p.shape
        Co = KC // K
        dev = p.device
        lib = native.lib()
        gy, gscale, gshift = _f32c(gy), _f32c(gscale), _f32c(gshift)
        A0 = B0 = dgamma = dbeta = None
        if want_bn:
            dgamma, dbeta, A0, B0 = _bn_coef(ctx.bn, gscale, gshift, mean, var, gamma, eps, count, Co, Co)
        if gy is None:
            gy = torch.zeros_like(y)
        dp = torch.empty_like(p)
        if per_sample:
            # channel-major per-(n, c) pieces: the ordered sum over the channels is one dsgcn_colsum over dim 0
            dpiece = torch.empty((Co, n, K, V, V), device=dev, dtype=torch.float32)
            dstr = (K * V * V, V * V, n * K * V * V)
        elif shared:
            rows = lib.dsgcn_aggsum_bwd_piece_rows(n, K, Co, T, V)       # per-wave pieces, or per-(n,c) ones if 0
            dpiece = torch.empty((rows or n * Co, K, V, V), device=dev, dtype=torch.float32)
            dstr = (Co * K * V * V, V * V, K * V * V)
        else:
            dpiece = torch.empty_like(adj)
            dstr = astr
        rc = lib.dsgcn_aggsum_bwd(_ptr(p), _ptr(adj), *astr, _ptr(gy), _ptr(y), _ptr(A0), _ptr(B0), _ptr(dp),
                                  _ptr(dpiece), *dstr, n, K, Co, T, V, _stream())
        native.check(rc, 'dsgcn_aggsum_bwd')
        if shared:      # (a parameter's gradient when A is used as is — unit_gcn 'init': its sum joins the deferred launches)
            dadj = param_colsum(dpiece, ctx.defer_ok)
        else:
            dadj = colsum(dpiece) if per_sample else dpiece
        if dgamma is not None:
            dgamma = dgamma if gamma is not None else None
            dbeta = dbeta if has_beta else None
        return dp, dadj, None, dgamma, dbeta, None, None, None, None


def aggregate_sum(p, adj, K, gamma=None, beta=None, eps=1e-5, want_bn=False, per_sample=False):
    """y[n,c,t,w] = sum_k sum_u p[n,k*Co+c,t,u] * adj_k[u,w]; adj (K,V,V) shared (ST-GCN unit_gcn) or (n,K*Co,V,V)
    per sample and channel (CTR-GCN), or — per_sample — (n,K,V,V) per sample shared by the channels (AAGCN), plus the
    train-mode BN of y as a deferred affine.  -> (y, scale, shift, mean, var)"""
    bn = BNCtx() if want_bn else None
    out = _AggSum.apply(p, adj, int(K), gamma, beta, float(eps), bool(want_bn), bool(per_sample), bn)
    if want_bn:
        n, KC, T, V = p.shape
        _bn_attach(out[1], bn, out[3], out[4], gamma, eps, float(n * T * V), KC // int(K))
    return out


# ---------------------------------------------------------------------------------------------
# AAGCN: embedding Gram and attention gates (gcn.py:431-437, 447-459)
# ---------------------------------------------------------------------------------------------

class _Gram(torch.autograd.Function):
    """G[n,u,w] = sum_{c,t} a[n,c,t,u] * b[n,c,t,w] for a, b (n, C, T, V): K-A''s backward product (the per-plane
    P^T dY pieces of dsgcn_aggsum_bwd, summed over the channels by an ordered dsgcn_colsum); its own backward is two K-A'
    forward launches with the per-sample V x V factor shared by the channels."""

    @staticmethod
    def forward(ctx, a, b):
        _require_cuda(a, b)
        a, b = _f32c(a), _f32c(b)
        n, C, T, V = a.shape
        dev = a.device
        zero = torch.zeros((n, V, V), device=dev, dtype=torch.float32)
        scratch = torch.empty_like(a)                                  # the launch also writes d p (unused here)
        piece = torch.empty((C, n, V, V), device=dev, dtype=torch.float32)
        rc = native.lib().dsgcn_aggsum_bwd(_ptr(a), _ptr(zero), V * V, 0, 0, _ptr(b), None, None, None, _ptr(scratch),
                                           _ptr(piece), V * V, 0, n * V * V, n, 1, C, T, V, _stream())
        native.check(rc, 'dsgcn_aggsum_bwd')
        ctx.save_for_backward(a, b)
        return colsum(piece)

    @staticmethod
    def backward(ctx, dG):
        a, b = ctx.saved_tensors
        n, C, T, V = a.shape
        dG = _f32c(dG)
        dGt = dG.transpose(1, 2).contiguous()
        lib = native.lib()
        da, db = torch.empty_like(a), torch.empty_like(b)
        # da[t,u] = sum_w b[t,w] dG[u,w] = (b . dG^T);  db[t,w] = sum_u a[t,u] dG[u,w]
        native.check(lib.dsgcn_aggsum_fwd(_ptr(b), _ptr(dGt), V * V, 0, 0, _ptr(da), None, n, 1, C, T, V, _stream()),
                     'dsgcn_aggsum_fwd')
        native.check(lib.dsgcn_aggsum_fwd(_ptr(a), _ptr(dG), V * V, 0, 0, _ptr(db), None, n, 1, C, T, V, _stream()),
                     'dsgcn_aggsum_fwd')
        return da, db


def gram(a, b):
    return _Gram.apply(a, b)


class _Gate(torch.autograd.Function):
    """out = y * (1 + g) with g broadcast per joint / frame / channel (mode 0 / 1 / 2) and the mean the next gate needs
    (rmode 1: over joints, 2: over the plane) in the same pass (csrc/aagcn.hip)."""

    @staticmethod
    def forward(ctx, y, g, mode, rmode):
        _require_cuda(y, g)
        y, g = _f32c(y), _f32c(g)
        n, C, T, V = y.shape
        assert g.shape == ((n, V), (n, T), (n, C))[mode], (g.shape, y.shape, mode)
        out = torch.empty_like(y)
        rout = None
        if rmode:
            rout = torch.empty((n, C, T) if rmode == 1 else (n, C), device=y.device, dtype=torch.float32)
        rc = native.lib().dsgcn_gate_fwd(_ptr(y), _ptr(g), int(mode), _ptr(out), _ptr(rout), int(rmode), n, C, T, V, _stream())
        native.check(rc, 'dsgcn_gate_fwd')
        ctx.save_for_backward(y, g)
        ctx.cfg = (int(mode), int(rmode))
        ctx.set_materialize_grads(False)
        return out, rout

    @staticmethod
    def backward(ctx, gout, drout):
        y, g = ctx.saved_tensors
        mode, rmode = ctx.cfg
        n, C, T, V = y.shape
        gout, drout = _f32c(gout), _f32c(drout)
        if gout is None:
            gout = torch.zeros_like(y)
        dy = torch.empty_like(y)
        dgp = torch.empty((n, C, V) if mode == 0 else ((n, C, T) if mode == 1 else (n, C)), device=y.device,
                          dtype=torch.float32)
        rc = native.lib().dsgcn_gate_bwd(_ptr(y), _ptr(g), mode, _ptr(gout), _ptr(drout), rmode, _ptr(dy), _ptr(dgp), n, C,
                                         T, V, _stream())
        native.check(rc, 'dsgcn_gate_bwd')
        dg = dgp if mode == 2 else dgp.sum(1)              # (n, C, .) -> (n, .): KB-sized
        return dy, dg, None, None


def gate(y, g, mode, rmode):
    return _Gate.apply(y, g, int(mode), int(rmode))


# ---------------------------------------------------------------------------------------------
# CTR-GCN channel-wise topology (gcn.py:634-666)
# ---------------------------------------------------------------------------------------------

class _TanhDiff(torch.autograd.Function):
    """proj (n, 2*K*R, V) -> K tensors d_k (n, R, V, V) = tanh(x1_k[..., u] - x2_k[..., v]) (slices of one buffer).  One
    output per subset: their gradients arrive one by one and go to the kernel as they are — indexing a stacked output made
    autograd build the stacked gradient with a fill, a copy and an add per subset (nine launches per CTR-GCN block)."""

    @staticmethod
    def forward(ctx, proj, K, R):
        _require_cuda(proj)
        proj = _f32c(proj)
        n, _, V = proj.shape
        d = torch.empty((K, n, R, V, V), device=proj.device, dtype=torch.float32)
        native.check(native.lib().dsgcn_tanhdiff_fwd(_ptr(proj), _ptr(d), n, K, R, V, _stream()), 'dsgcn_tanhdiff_fwd')
        ctx.save_for_backward(d)
        ctx.dims = (n, K, R, V)
        ctx.set_materialize_grads(False)
        return tuple(d[k] for k in range(K))

    @staticmethod
    def backward(ctx, *dds):
        d, = ctx.saved_tensors
        n, K, R, V = ctx.dims
        if all(g is None for g in dds):
            return None, None, None
        dds = [_f32c(g) for g in dds]
        dproj = torch.empty((n, 2 * K * R, V), device=d.device, dtype=torch.float32)
        native.check(native.lib().dsgcn_tanhdiff_bwd_k(_ptr(d), _ptr_array(dds), _ptr(dproj), n, K, R, V, _stream()),
                     'dsgcn_tanhdiff_bwd_k')
        return dproj, None, None


class _TanhDiffAug(torch.autograd.Function):
    """proj (n, 2*K*R, V), A (K, V, V) -> K tensors (n, R + 2, V, V) = [tanh(x1_k - x2_k) | A[k] | 1]: the operands of the
    one-conv form of the classic refinement (ctr_topology).  Backward: the first R channels' gradients -> dproj, the A
    channel's -> dA (its sum over the samples rides in the deferred parameter sums)."""

    @staticmethod
    def forward(ctx, proj, A, K, R):
        _require_cuda(proj, A)
        proj, A = _f32c(proj), _f32c(A)
        n, _, V = proj.shape
        d = torch.empty((K, n, R + 2, V, V), device=proj.device, dtype=torch.float32)
        native.check(native.lib().dsgcn_tanhdiff_aug_fwd(_ptr(proj), _ptr(A), _ptr(d), n, K, R, V, _stream()),
                     'dsgcn_tanhdiff_aug_fwd')
        ctx.save_for_backward(d)
        ctx.dims = (n, K, R, V)
        ctx.defer_ok = _leafish(A)
        ctx.set_materialize_grads(False)
        return tuple(d[k] for k in range(K))

    @staticmethod
    def backward(ctx, *dds):
        d, = ctx.saved_tensors
        n, K, R, V = ctx.dims
        if all(g is None for g in dds):
            return None, None, None, None
        dds = [_f32c(g) for g in dds]
        dproj = torch.empty((n, 2 * K * R, V), device=d.device, dtype=torch.float32)
        dAp = torch.empty((n, K * V * V), device=d.device, dtype=torch.float32)
        rc = native.lib().dsgcn_tanhdiff_aug_bwd(_ptr(d), _ptr_array(dds), _ptr(dproj), _ptr(dAp), n, K, R, V, _stream())
        native.check(rc, 'dsgcn_tanhdiff_aug_bwd')
        return dproj, param_colsum(dAp, ctx.defer_ok).view(K, V, V), None, None


# The augmented operands depend on parameters only: inside a step the first unit that asks rebuilds the operands of ALL
# units seen in the previous step with one dsgcn_ctr_wprep_multi launch (_StepBuilt: 10 dependent ~5 us launches per
# CTR-GCN step -> 1), and the finishing launches of the backward ride in one dsgcn_ctr_wfin_multi behind the deferred
# column sums.
CTR_PREP_BATCH = _os.environ.get('DSGCN_CTR_PREP_BATCH', '1') == '1'
CTR_FIN_BATCH = _os.environ.get('DSGCN_CTR_FIN_BATCH', '1') == '1'
_ctr_fin_queue = []


def _ctr_prep_launch(todo):
    tab = (native.CtrPrepJob * len(todo))()
    for rec, job in zip(tab, todo):
        K, Co, R = job['dims']
        alpha, wb = job['src'][0], job['src'][1:]
        for k in range(K):
            rec.w[k] = wb[k].data_ptr()
            rec.b[k] = wb[K + k].data_ptr() if wb[K + k] is not None else None
        rec.alpha, rec.wout, rec.sh = alpha.data_ptr(), job['wout'].data_ptr(), job['sh'].data_ptr()
        rec.K, rec.Co, rec.R = K, Co, R
    native.check(native.lib().dsgcn_ctr_wprep_multi(tab, len(todo), _stream()), 'dsgcn_ctr_wprep_multi')


_ctr_prep_cache = _StepBuilt(_ctr_prep_launch, lambda: CTR_PREP_BATCH)


def _ctr_prep_operands(alpha, w, b):
    """-> (wout (K, Co, R + 2), sh (K, 2, R + 2)) for this unit, built now or taken from this step's batched launch."""
    K = len(w)
    Co, R = w[0].shape
    dev = alpha.device
    job = _ctr_prep_cache.get((Co, R), [alpha, *w, *b], lambda: dict(
        dims=(K, Co, R), wout=torch.empty((K, Co, R + 2), device=dev, dtype=torch.float32),
        sh=torch.empty((K, 2, R + 2), device=dev, dtype=torch.float32)))
    return job['wout'], job['sh']


def _ctr_fin_flush():
    """One dsgcn_ctr_wfin_multi for every finishing launch queued since the last flush."""
    queued = list(_ctr_fin_queue)
    _ctr_fin_queue.clear()
    if queued:
        _ctr_fin_launch(queued)


def _ctr_fin_launch(queued):
    tab = (native.CtrFinJob * len(queued))()
    for rec, (dwp, ds, stride, outs, dalpha, K, Co, R) in zip(tab, queued):
        for k in range(K):
            rec.dwp[k] = dwp[k].data_ptr() if dwp[k] is not None else None
            rec.ds[k] = ds[k].data_ptr() if ds[k] is not None else None
            rec.out[k] = outs[k].data_ptr()
        rec.dalpha, rec.K, rec.Co, rec.R, rec.ds_stride = dalpha.data_ptr(), K, Co, R, int(stride)
    native.check(native.lib().dsgcn_ctr_wfin_multi(tab, len(queued), _stream()), 'dsgcn_ctr_wfin_multi')


class _CtrWPrep(torch.autograd.Function):
    """(alpha (1), W_0 .. W_{K-1} (Co, R), b_0 .. b_{K-1} (Co) | None) -> K augmented weights [W_k | 1 | b_k] (Co, R + 2), K
    input scales [alpha x R, 1, alpha] and K zero shifts (one launch per step for all units: _ctr_prep_operands; one launch
    back for all units: dW_k, db_k, dalpha)."""

    @staticmethod
    def forward(ctx, use, alpha, *wb):
        """use: the ``_leafish(alpha, *wb)`` of the caller — the same object rides on the outputs as ``_dsgcn_sink`` so that
        the consuming convs and this node's backward take ONE decision about deferring."""
        K = len(wb) // 2
        w = [_f32c(t) for t in wb[:K]]
        b = [_f32c(t) for t in wb[K:]]
        _require_cuda(alpha, *w)
        alpha = _f32c(alpha)
        Co, R = w[0].shape
        wout, sh = _ctr_prep_operands(alpha, w, b)
        ctx.dims = (K, Co, R, tuple(t is not None for t in b), alpha.shape)
        ctx.defer_ok = use
        shifts = tuple(sh[k, 1] for k in range(K))
        ctx.mark_non_differentiable(*shifts)
        ctx.set_materialize_grads(False)
        return tuple(wout[k] for k in range(K)) + tuple(sh[k, 0] for k in range(K)) + shifts

    @staticmethod
    def backward(ctx, *grads):
        K, Co, R, has_b, ashape = ctx.dims
        dwp = [_f32c(g) for g in grads[:K]]
        ds = list(grads[K:2 * K])                   # (R + 2,) each: contiguous, or column 0 of deferred (R + 2, 3) sums
        strides = {g.stride(0) for g in ds if g is not None}
        if len(strides) > 1:
            ds = [None if g is None else g.contiguous() for g in ds]
            strides = {1}
        live = next((g for g in dwp + ds if g is not None), None)
        if live is None:
            return (None,) * (2 + 2 * K)
        out = torch.empty((K, Co * R + Co), device=live.device, dtype=torch.float32)
        dalpha = torch.empty(1, device=live.device, dtype=torch.float32)
        outs = [out[k] for k in range(K)]
        rec = (dwp, ds, strides.pop() if strides else 1, outs, dalpha, K, Co, R)
        if _deferred is not None and ctx.defer_ok and not CTR_FIN_BATCH:
            _post_flush.append(lambda rec=rec: _ctr_fin_launch([rec]))
        elif _deferred is not None and ctx.defer_ok:
            _ctr_fin_queue.append(rec)                  # its inputs are deferred sums: filled by the flush, which then
            if _ctr_fin_flush not in _post_flush:       # runs ONE finishing launch for every unit queued by then
                _post_flush.append(_ctr_fin_flush)
        else:
            _ctr_fin_launch([rec])
        return (None, dalpha.view(ashape), *[o[:Co * R].view(Co, R) for o in outs],
                *[(o[Co * R:] if hb else None) for o, hb in zip(outs, has_b)])


class _JoinSlices(torch.autograd.Function):
    """K tensors that are the slices buf[0..K-1] of one buffer -> the buffer itself (no copy); the gradient goes back as its
    K slices.  (pwconv(out=OutSlot(buf[k])) fills such slices.)"""

    @staticmethod
    def forward(ctx, slot, *parts):
        buf = slot.t
        step = buf[0].numel() * buf.element_size()
        if len(parts) != buf.shape[0] or any(p.data_ptr() != buf.data_ptr() + k * step or p.shape != buf.shape[1:]
                                              for k, p in enumerate(parts)):
            raise ValueError('_JoinSlices: the parts are not the slices of the buffer')
        return buf.view(buf.shape)

    @staticmethod
    def backward(ctx, g):
        g = _f32c(g)
        return (None, *[g[k] for k in range(g.shape[0])])


CTR_ONE_CONV = _os.environ.get('DSGCN_CTR_ONE_CONV', '1') != '0'     # classic refinement: conv4 + affine as one conv (A/B)


class _CtrAffine(torch.autograd.Function):
    """Ahat (n, K*Co, V, V) = alpha_k * S_k + A[k] (+ beta_k * G[:, k])  for the K per-subset conv4 outputs S_k
    (n, Co, V, V).  alpha: 1 element (classic CTR-GCN: shared) or K elements (CTRHGC: per subset); G (n, K, V, V) with
    beta (K): the optional Gram term of CTRHGC's ``ada`` branch."""

    @staticmethod
    def forward(ctx, alpha, A, beta, G, *S):
        _require_cuda(A, *S)
        alpha, A, beta, G = _f32c(alpha), _f32c(A), _f32c(beta), _f32c(G)
        S = [_f32c(t) for t in S]
        K = len(S)
        n, Co, V, _ = S[0].shape
        astride = 1 if alpha.numel() == K and K > 1 else 0
        assert alpha.numel() in (1, K)
        ahat = torch.empty((n, K * Co, V, V), device=A.device, dtype=torch.float32)
        rc = native.lib().dsgcn_ctr_affine_fwd(_ptr_array(S), _ptr(alpha), astride, _ptr(A), _ptr(beta), _ptr(G),
                                               _ptr(ahat), n, K, Co, V, _stream())
        native.check(rc, 'dsgcn_ctr_affine_fwd')
        ctx.save_for_backward(alpha, beta, G, *S)
        ctx.astride = astride
        ctx.defer_ok = _leafish(alpha, A) if G is None else False
        return ahat

    @staticmethod
    def backward(ctx, dahat):
        alpha, beta, G, *S = ctx.saved_tensors
        K = len(S)
        n, Co, V, _ = S[0].shape
        dahat = _f32c(dahat)
        dS = [torch.empty_like(t) for t in S]
        prow = torch.empty((4 * n, K * V * V + K), device=dahat.device, dtype=torch.float32)   # 4 channel slices
        rc = native.lib().dsgcn_ctr_affine_bwd(_ptr_array(S), _ptr(alpha), ctx.astride, _ptr(dahat), _ptr_array(dS),
                                               _ptr(prow), n, K, Co, V, _stream())
        native.check(rc, 'dsgcn_ctr_affine_bwd')
        red = param_colsum(prow, ctx.defer_ok)          # (A and alpha are parameters: both sums ride in the flush)
        dA = red[:K * V * V].view(K, V, V)
        ds_k = red[K * V * V:]                                      # sum dAhat * S per subset
        dalpha = ds_k.view_as(alpha) if ctx.astride else param_colsum(ds_k.reshape(K, 1), ctx.defer_ok, 1).view_as(alpha)
        dbeta = dG = None
        if G is not None:
            # per-sample sum over channels of dAhat (the 4 channel slices of each sample added): KB-sized host-side algebra
            sumc = prow.view(n, 4, -1)[:, :, :K * V * V].sum(1).view(n, K, V, V)
            dG = sumc * beta.view(1, K, 1, 1)
            dbeta = (sumc * G).sum((0, 2, 3))
        return (dalpha, dA, dbeta, dG, *dS)


class _EdgeSelect(torch.autograd.Function):
    """x (n, E*R, V, V) -> (n, R, V, V): joint pair (u,v) keeps channel eps(u,v)*R + r (CTRHGC edge attention)."""

    @staticmethod
    def forward(ctx, x, edge_type, R):
        _require_cuda(x)
        x = _f32c(x)
        n, ER, V, _ = x.shape
        E = ER // R
        out = torch.empty((n, R, V, V), device=x.device, dtype=torch.float32)
        rc = native.lib().dsgcn_edge_select_fwd(_ptr(x), _ptr(edge_type), _ptr(out), n, R, E, V, _stream())
        native.check(rc, 'dsgcn_edge_select_fwd')
        ctx.save_for_backward(edge_type)
        ctx.dims = (n, R, E, V)
        return out

    @staticmethod
    def backward(ctx, dout):
        edge_type, = ctx.saved_tensors
        n, R, E, V = ctx.dims
        dout = _f32c(dout)
        din = torch.empty((n, E * R, V, V), device=dout.device, dtype=torch.float32)
        rc = native.lib().dsgcn_edge_select_bwd(_ptr(dout), _ptr(edge_type), _ptr(din), n, R, E, V, _stream())
        native.check(rc, 'dsgcn_edge_select_bwd')
        return din, None, None


def ctr_topology(xbar, w1, b1, w2, b2, w4, b4, alpha, A, beta=None, edge=None, subset_major=False):
    """CTR-GCN refined topology.  xbar (n,Ci,V) = mean_T x; w1,w2 (K*R,Ci) / b1,b2 (K*R): conv1/conv2 of the K
    subsets stacked (their mean over T commutes with the 1x1 conv); w4[k] (Co,R), b4[k] (Co); A (K,V,V).
    alpha (1): classic unit_ctrgcn;  alpha (K) [+ beta (K)]: unit_ctrhgcn — per-subset scale and the Gram term
    beta_k * x1_k^T x2_k;  edge = {k: (w_edge (E*R,R), b_edge (E*R), edge_type (V*V) int32)}: subsets whose tanh-difference
    first passes the edge-typed attention conv + select (gcn.py:737-745).
    -> Ahat (n, K*Co, V, V) = alpha_k * conv4_k(sel_k(tanh(x1_k[u] - x2_k[v]))) + A[k] (+ beta_k G_k).
    subset_major: the caller also takes Ahat as (K, n, Co, V, V) (``aggregate_sum`` does, by strides); the classic unit
    then comes through the one-conv form below and in that layout."""
    n, Ci, V = xbar.shape
    K = A.shape[0]
    R = w1.shape[0] // K
    proj = pwconv(xbar.unsqueeze(2), None, None, None, False, cat_rows([w1, w2]), cat_rows([b1, b2]), 1,
                  False)[0].view(n, 2 * K * R, V)
    if subset_major and CTR_ONE_CONV and beta is None and not edge and alpha.numel() == 1 and K <= 4:
        # classic unit: Ahat_k = alpha * conv4_k(d_k) + A[k] as ONE conv per subset over [d_k | A[k] | 1] (include/dsgcn.h,
        # dsgcn_tanhdiff_aug_fwd) writing straight into its slice of Ahat (K, n, Co, V, V) — the layout aggregate_sum
        # takes by strides
        d = _TanhDiffAug.apply(proj, A, K, R)
        use = _leafish(alpha, *w4, *b4)
        prep = _CtrWPrep.apply(use, alpha, *w4, *b4)
        for t in prep[:2 * K]:
            t._dsgcn_sink = use                 # their gradients are only read by _CtrWPrep's finishing launch
        Co = w4[0].shape[0]
        buf = torch.empty((K, n, Co, V, V), device=xbar.device, dtype=torch.float32)
        parts = pwconv_group([d[k] for k in range(K)], [(prep[K + k], prep[2 * K + k]) for k in range(K)],
                             [prep[k] for k in range(K)], [OutSlot(buf[k]) for k in range(K)])
        return _JoinSlices.apply(OutSlot(buf), *parts)
    d = _TanhDiff.apply(proj, K, R)
    S = []
    for k in range(K):
        dk = d[k]
        if edge and k in edge:
            we, be, et = edge[k]
            dk = _EdgeSelect.apply(pwconv(dk, None, None, None, False, we, be, 1, False)[0], et, R)
        S.append(pwconv(dk, None, None, None, False, w4[k], b4[k], 1, False)[0])
    G = None
    if beta is not None:      # Gram of the mean-pooled projections per subset (gcn.py:826-835): kernels.gram with one-frame planes
        x1 = proj[:, :K * R].reshape(n * K, R, 1, V)
        x2 = proj[:, K * R:].reshape(n * K, R, 1, V)
        G = ops().gram(x1, x2).view(n, K, V, V)
    return _CtrAffine.apply(alpha, A, beta, G, *S)


class _Tee3(torch.autograd.Function):
    """x -> three aliases of x, one per consumer; backward sums their gradients in ONE launch (dsgcn_add3) instead of
    autograd's pairwise accumulation (two launches, six plane accesses instead of four)."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        return x.view_as(x), x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb, gc):
        gs = [_f32c(g) for g in (ga, gb, gc) if g is not None]
        if not gs:
            return None
        if len(gs) == 1:
            return gs[0]
        out = torch.empty_like(gs[0])
        rc = native.lib().dsgcn_add3(_ptr(gs[0]), _ptr(gs[1]), _ptr(gs[2]) if len(gs) > 2 else None, _ptr(out),
                                     out.numel(), _stream())
        native.check(rc, 'dsgcn_add3')
        return out


def tee3(x):
    """Three aliases of a block input for its three consumers (see _Tee3)."""
    _require_cuda(x)
    if not x.requires_grad:
        return x, x, x
    return _Tee3.apply(x)


class Prestrided:
    """The even frames of a block output, handed to the next block's stride-2 residual conv as a tensor of their own
    (_FuseOut, tee = 2).  An explicit wrapper rather than an attribute on the tensor: anything that makes a new tensor
    object (detach, alias, a subclass's forward) would drop an attribute silently and the conv would stride the already
    halved tensor.  ``frames``: the frame count of the full-rate tensor the kept frames came from."""
    __slots__ = ('x', 'stride', 'frames')

    def __init__(self, x, stride, frames):
        if x.shape[2] != (frames + stride - 1) // stride:
            raise ValueError(f'Prestrided: {x.shape[2]} frames are not every {stride}-th of {frames}')
        self.x, self.stride, self.frames = x, int(stride), int(frames)


FUSE_OUT_LDS_BYTES = 64 * 1024        # k_fuse_out_fwd keeps the whole (T, V) plane in LDS for the time mean / even frames


def prestrided_fits(T, V):
    """Can fuse_out(tee=2) take planes of this size?  (longer planes go through strided_frames as before)"""
    return T * V * 4 <= FUSE_OUT_LDS_BYTES


# ---- dropout inside fuse_out (csrc/dropout.h) --------------------------------------------------------------------------
# The reference's temporal units end in nn.Dropout(p, inplace=True) behind their BatchNorm (tcn.py:30,33); vanilla ST-GCN
# trains with p = 0.5.  Here the mask is never a tensor: fuse_out multiplies its first term by keep / (1 - p) from a
# counter-based generator keyed by (seed, training step, call), and its backward regenerates the same multipliers.
#   seed  = torch.initial_seed() (torch.manual_seed governs it);
#   call  = a host counter, one value per fuse_out call (distinct masks per layer and per eager call);
#   step  = a DEVICE counter (a captured graph freezes `call`: the replays differ through it), advanced once per training
#           step by TrainEngine (dropout_step_advance) — never between a forward and its backward.
_drop_state = {'call': 0, 'step': {}, 'used': False}


def _dropout_record(p, device):
    """-> (native.Dropout, keep-alive tuple) for one fuse_out call with drop probability p"""
    st = _drop_state
    if device not in st['step']:
        st['step'][device] = torch.zeros((), dtype=torch.int64, device=device)
    st['call'] = (st['call'] + 1) & 0x7fffffff
    st['used'] = True
    step = st['step'][device]
    return native.Dropout(step.data_ptr(), int(torch.initial_seed()) & 0xffffffffffffffff, st['call'], float(p)), step


def dropout_step_advance():
    """End of a training step (TrainEngine): the next step's masks differ.  One tiny launch, and only in models that use
    fused dropout."""
    if _drop_state['used']:
        for step in _drop_state['step'].values():
            step.add_(1)


def dropout_mask(numel, rec):
    """The multipliers (numel,) a tensor of numel elements gets under record ``rec`` (tests)."""
    dev = next(iter(_drop_state['step']))
    out = torch.empty(numel, device=dev, dtype=torch.float32)
    native.check(native.lib().dsgcn_dropout_mask(_ptr(out), numel, _ct.byref(rec), _stream()), 'dsgcn_dropout_mask')
    return out


class _FuseOut(torch.autograd.Function):
    """-> (out, out', out'', xbar): with ``tee`` the output comes as three aliases, one per consumer in the next block
    (spatial unit, its residual operand, block residual), and the backward sums their gradients while loading them —
    autograd's own accumulation would be two more launches and a materialised sum (the round-2 dsgcn_add3 pass).
    ``tee`` = 2: the third output is not an alias but the EVEN FRAMES of out as a tensor of their own — what the 1x1
    residual conv of a stride-2 block reads (dgstgcn.py:35-40); its gradient comes back in that shape and is added on
    the even frames inside the same backward launch (no strided-copy launch, no scatter into a zero-filled tensor)."""

    @staticmethod
    def forward(ctx, x1, s1, h1, x2, s2, h2, relu, xbar_ld, tee, dropout=0.0):
        _require_cuda(x1)
        x1, s1, h1, x2, s2, h2 = [_f32c(t) for t in (x1, s1, h1, x2, s2, h2)]
        n, C, T, V = x1.shape
        if x2 is not None and x2.shape != x1.shape:
            raise ValueError(f'fuse_out: the two terms differ in shape: {tuple(x1.shape)} vs {tuple(x2.shape)}')
        out = torch.empty_like(x1)
        out_s2 = torch.empty((n, C, (T + 1) // 2, V), device=x1.device, dtype=torch.float32) if tee == 2 else None
        xbar = torch.empty((n, C, xbar_ld), device=x1.device, dtype=torch.float32) if xbar_ld else None
        ctx.drop = _dropout_record(dropout, x1.device) if dropout > 0 else None
        rc = native.lib().dsgcn_fuse_out_fwd_drop(_ptr(x1), _ptr(s1), _ptr(h1), _ptr(x2), _ptr(s2), _ptr(h2), int(relu),
                                                  _ptr(out), _ptr(out_s2), _ptr(xbar), None, n, C, T, V, int(xbar_ld),
                                                  _ct.byref(ctx.drop[0]) if ctx.drop else None, _stream())
        native.check(rc, 'dsgcn_fuse_out_fwd_drop')
        ctx.save_for_backward(x1, s1, h1, x2, s2, h2)
        ctx.relu = int(relu)
        ctx.xbar_ld = int(xbar_ld)
        ctx.tee = int(tee)
        ctx.bn1, ctx.bn2 = _bn_of(s1), _bn_of(s2)
        ctx.set_materialize_grads(False)
        if tee == 2:
            return out, out.view_as(out), out_s2, xbar
        if tee:
            return out, out.view_as(out), out.view_as(out), xbar
        return out, None, None, xbar

    @staticmethod
    def backward(ctx, dout, dout2, dout3, dxbar):
        x1, s1, h1, x2, s2, h2 = ctx.saved_tensors
        n, C, T, V = x1.shape
        stride3 = 1
        for g, frames in ((dout, T), (dout2, T), (dout3, (T + 1) // 2 if ctx.tee == 2 else T)):
            if g is not None and tuple(g.shape) != (n, C, frames, V):
                raise ValueError(f'fuse_out backward: gradient of shape {tuple(g.shape)}, expected {(n, C, frames, V)}')
        if ctx.tee == 2:
            # the third stream has the even-frame shape: it keeps its slot (the kernel needs a first stream to add it to)
            d3 = _f32c(dout3)
            full = [_f32c(g) for g in (dout, dout2) if g is not None]
            if d3 is not None and not full:
                full = [torch.zeros_like(x1)]
            douts = full + [None] * (2 - len(full)) + [d3]
            stride3 = 2 if d3 is not None else 1
        else:
            douts = [_f32c(g) for g in (dout, dout2, dout3) if g is not None]
            douts += [None] * (3 - len(douts))
        dxbar = _f32c(dxbar)
        dx1 = torch.empty_like(x1)
        dx2 = torch.empty_like(x2) if x2 is not None else None
        need_part = s1 is not None or s2 is not None
        part = torch.empty((n, C, 4), device=x1.device, dtype=torch.float32) if need_part else None
        rc = native.lib().dsgcn_fuse_out_bwd_drop(_ptr(x1), _ptr(s1), _ptr(h1), _ptr(x2), _ptr(s2), _ptr(h2), ctx.relu,
                                                  _ptr(douts[0]), _ptr(douts[1]), _ptr(douts[2]), stride3, _ptr(dxbar),
                                                  _ptr(dx1), _ptr(dx2), _ptr(part), n, C, T, V, ctx.xbar_ld or V,
                                                  _ct.byref(ctx.drop[0]) if ctx.drop else None, _stream())
        native.check(rc, 'dsgcn_fuse_out_bwd_drop')
        ds1 = dh1 = ds2 = dh2 = None
        if need_part and (s1 is None or ctx.bn1 is not None) and (s2 is None or ctx.bn2 is not None):
            _bn_feed_multi(([(ctx.bn1, part, 4, 0, 3)] if s1 is not None else []) +
                           ([(ctx.bn2, part, 4, 2, 1)] if s2 is not None else []))
        elif need_part:
            red = colsum(part, split_last=True)
            if s1 is not None:
                ds1, dh1 = red[0], red[3]
            if s2 is not None:
                ds2, dh2 = red[2], red[1]
        return dx1, ds1, dh1, dx2, ds2, dh2, None, None, None, None


class _FuseOutPool(torch.autograd.Function):
    """The block output as its plane means (n, C) only — the LAST block, whose activation nothing but the head's pooling
    reads (_FuseOut without the write; the backward spreads the pooled gradient while it loads its operands)."""

    @staticmethod
    def forward(ctx, x1, s1, h1, x2, s2, h2, relu, dropout=0.0):
        _require_cuda(x1)
        x1, s1, h1, x2, s2, h2 = [_f32c(t) for t in (x1, s1, h1, x2, s2, h2)]
        n, C, T, V = x1.shape
        if x2 is not None and x2.shape != x1.shape:
            raise ValueError(f'fuse_out_pool: the two terms differ in shape: {tuple(x1.shape)} vs {tuple(x2.shape)}')
        pm = torch.empty((n, C), device=x1.device, dtype=torch.float32)
        ctx.drop = _dropout_record(dropout, x1.device) if dropout > 0 else None
        rc = native.lib().dsgcn_fuse_out_fwd_drop(_ptr(x1), _ptr(s1), _ptr(h1), _ptr(x2), _ptr(s2), _ptr(h2), int(relu),
                                                  None, None, None, _ptr(pm), n, C, T, V, V,
                                                  _ct.byref(ctx.drop[0]) if ctx.drop else None, _stream())
        native.check(rc, 'dsgcn_fuse_out_fwd_drop')
        ctx.save_for_backward(x1, s1, h1, x2, s2, h2)
        ctx.relu = int(relu)
        ctx.bn1, ctx.bn2 = _bn_of(s1), _bn_of(s2)
        return pm

    @staticmethod
    def backward(ctx, dpm):
        x1, s1, h1, x2, s2, h2 = ctx.saved_tensors
        n, C, T, V = x1.shape
        dpm = _f32c(dpm)
        dx1 = torch.empty_like(x1)
        dx2 = torch.empty_like(x2) if x2 is not None else None
        need_part = s1 is not None or s2 is not None
        part = torch.empty((n, C, 4), device=x1.device, dtype=torch.float32) if need_part else None
        rc = native.lib().dsgcn_fuse_out_bwd_drop(_ptr(x1), _ptr(s1), _ptr(h1), _ptr(x2), _ptr(s2), _ptr(h2), ctx.relu,
                                                  _ptr(dpm), None, None, 3, None, _ptr(dx1), _ptr(dx2), _ptr(part), n, C, T, V,
                                                  V, _ct.byref(ctx.drop[0]) if ctx.drop else None, _stream())
        native.check(rc, 'dsgcn_fuse_out_bwd_drop')
        ds1 = dh1 = ds2 = dh2 = None
        if need_part and (s1 is None or ctx.bn1 is not None) and (s2 is None or ctx.bn2 is not None):
            _bn_feed_multi(([(ctx.bn1, part, 4, 0, 3)] if s1 is not None else []) +
                           ([(ctx.bn2, part, 4, 2, 1)] if s2 is not None else []))
        elif need_part:
            red = colsum(part, split_last=True)
            if s1 is not None:
                ds1, dh1 = red[0], red[3]
            if s2 is not None:
                ds2, dh2 = red[2], red[1]
        return dx1, ds1, dh1, dx2, ds2, dh2, None, None


DROPOUT_FUSED = _os.environ.get('DSGCN_DROPOUT_FUSED', '1') != '0'     # '0': materialise + torch dropout (the round-5 path)


def fuse_out_pool(x1, a1, x2, a2, relu, dropout=0.0):
    """mean over (T, V) of ``fuse_out(x1, a1, x2, a2, relu)`` -> (n, C), without materialising the activation."""
    s1, h1 = a1 if a1 is not None else (None, None)
    s2, h2 = a2 if a2 is not None else (None, None)
    return _FuseOutPool.apply(x1, s1, h1, x2, s2, h2, int(relu), float(dropout))


def fuse_out(x1, a1, x2, a2, relu, want_tmean=False, tee=False, dropout=0.0):
    """relu: bool, or int flags — bit 0 the outer ReLU, bit 1 a ReLU on the first term before the add.
    want_tmean: False / True (time mean (n, C, V)) / an int ld >= V (time mean with the joint row zero-padded to ld: the
    layout `dynadj` consumes directly).
    tee: return the output as a tuple of three aliases (see _FuseOut) for the next block's three reads; tee = 2: the third
    one is the even-frame tensor (n, C, ceil(T/2), V) wrapped as ``Prestrided`` for the stride-2 residual conv.
    dropout: drop probability of the FIRST term (after its own ReLU, before the second term is added): the temporal unit's
    nn.Dropout (tcn.py:30,33) without a mask tensor or a pass of its own."""
    s1, h1 = a1 if a1 is not None else (None, None)
    s2, h2 = a2 if a2 is not None else (None, None)
    ld = 0 if not want_tmean else (x1.shape[-1] if want_tmean is True else int(want_tmean))
    o1, o2, o3, xbar = _FuseOut.apply(x1, s1, h1, x2, s2, h2, int(relu), ld, int(tee), float(dropout))
    if int(tee) == 2:
        o3 = Prestrided(o3, 2, x1.shape[2])
    return ((o1, o2, o3) if tee else o1), xbar


# ---------------------------------------------------------------------------------------------
# the small ends of the step (csrc/head.hip)
# ---------------------------------------------------------------------------------------------
class _HeadLoss(torch.autograd.Function):
    """(feat (N*M, C) per-person plane means, weight (K, C), bias (K), label (N) int64) -> (loss scalar, acc (2) fp64,
    score (N, K)); only the loss is differentiable.  Two launches forward, one backward."""

    @staticmethod
    def forward(ctx, feat, weight, bias, label, M, loss_weight):
        _require_cuda(feat, weight, label)
        feat, weight, bias = _f32c(feat), _f32c(weight), _f32c(bias)
        if label.dtype != torch.int64 or not label.is_contiguous():
            label = label.to(torch.int64).contiguous()
        R, C = feat.shape
        K = weight.shape[0]
        if R % M or weight.shape[1] != C or label.numel() * M != R:
            raise ValueError(f'head_loss: feat {tuple(feat.shape)}, weight {tuple(weight.shape)}, {label.numel()} labels, '
                             f'{M} persons do not fit together')
        N = R // M
        dev = feat.device
        pooled = torch.empty((N, C), device=dev, dtype=torch.float32)
        score = torch.empty((N, K), device=dev, dtype=torch.float32)
        prob = torch.empty((N, K), device=dev, dtype=torch.float32)
        clip = torch.empty((N, 3), device=dev, dtype=torch.float32)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        acc = torch.empty(2, device=dev, dtype=torch.float64)
        rc = native.lib().dsgcn_head_loss_fwd(_ptr(feat), _ptr(weight), _ptr(bias), _ptr(label), N, M, C, K,
                                              float(loss_weight), _ptr(pooled), _ptr(score), _ptr(prob), _ptr(clip),
                                              _ptr(loss), _ptr(acc), _stream())
        native.check(rc, 'dsgcn_head_loss_fwd')
        ctx.save_for_backward(prob, pooled, weight, label)
        ctx.dims = (N, M, C, K, float(loss_weight), bias is not None)
        ctx.mark_non_differentiable(acc, score)
        ctx.set_materialize_grads(False)            # (zero gradients for acc / score would be two fill launches)
        return loss, acc, score

    @staticmethod
    def backward(ctx, gloss, _gacc, _gscore):
        if gloss is None:
            return None, None, None, None, None, None
        prob, pooled, weight, label = ctx.saved_tensors
        N, M, C, K, lw, has_bias = ctx.dims
        dev = prob.device
        gloss = _f32c(gloss)
        dfeat = torch.empty((N * M, C), device=dev, dtype=torch.float32)
        dw = torch.empty((K, C), device=dev, dtype=torch.float32)
        db = torch.empty(K, device=dev, dtype=torch.float32)
        rc = native.lib().dsgcn_head_loss_bwd(_ptr(prob), _ptr(pooled), _ptr(weight), _ptr(label), _ptr(gloss), N, M, C, K,
                                              lw, _ptr(dfeat), _ptr(dw), _ptr(db), _stream())
        native.check(rc, 'dsgcn_head_loss_bwd')
        return dfeat, dw, (db if has_bias else None), None, None, None


def head_loss(feat, weight, bias, label, persons, loss_weight=1.0):
    """Person mean + Linear + softmax cross entropy (mean over the clips, times loss_weight) + top-1 / top-5 accuracy.
    -> (loss 0-dim fp32, acc (2,) fp64, score (N, K))."""
    return _HeadLoss.apply(feat, weight, bias, label, int(persons), float(loss_weight))


def bn_running_update(items):
    """items: (bn module, mean, var, count) of the BatchNorm layers a training forward went through (momentum form only):
    their buffers updated in ONE launch."""
    import ctypes as _c
    k = len(items)
    ptrs = lambda ts: (_c.c_void_p * k)(*[None if t is None else t.data_ptr() for t in ts])
    for bn, mean, var, _ in items:
        _require_cuda(bn.running_mean, mean, var)
        for t in (bn.running_mean, bn.running_var, mean, var):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != bn.num_features:
                raise ValueError('bn_running_update: fp32 contiguous per-channel vectors expected')
    nbt = [bn.num_batches_tracked for bn, _, _, _ in items]
    rc = native.lib().dsgcn_bn_running_multi(
        ptrs([bn.running_mean for bn, _, _, _ in items]), ptrs([bn.running_var for bn, _, _, _ in items]),
        ptrs([m for _, m, _, _ in items]), ptrs([v for _, _, v, _ in items]), ptrs(nbt),
        _int_array([bn.num_features for bn, _, _, _ in items]),
        (_c.c_float * k)(*[c / max(c - 1.0, 1.0) for _, _, _, c in items]),
        (_c.c_float * k)(*[float(bn.momentum) for bn, _, _, _ in items]), k, _stream())
    native.check(rc, 'dsgcn_bn_running_multi')


class _DataBN(torch.autograd.Function):
    """x (N, M, T, V, C) (no gradient), gamma, beta -> BatchNorm1d over the (v, c) / (m, v, c) channels, as (N*M, C, T, V)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, mvc, training):
        _require_cuda(x)
        x, gamma, beta = _f32c(x), _f32c(gamma), _f32c(beta)
        N, M, T, V, C = x.shape
        Ch = (M if mvc else 1) * V * C
        dev = x.device
        y = torch.empty((N * M, C, T, V), device=dev, dtype=torch.float32)
        track = bn.track_running_stats and bn.running_mean is not None
        save = torch.empty((2, Ch), device=dev, dtype=torch.float32) if training else None
        scratch = torch.empty(2 * N * M * V * C, device=dev, dtype=torch.float64) if training else None
        upd = training and track
        rc = native.lib().dsgcn_data_bn_fwd(
            _ptr(x), _ptr(gamma), _ptr(beta), _ptr(bn.running_mean) if (upd or not training) else None,
            _ptr(bn.running_var) if (upd or not training) else None, _ptr(bn.num_batches_tracked) if upd else None, _ptr(y),
            _ptr(save[0]) if training else None, _ptr(save[1]) if training else None, _ptr(scratch), N, M, T, V, C, int(mvc),
            int(training), float(bn.eps), float(bn.momentum if bn.momentum is not None else 0.0), _stream())
        native.check(rc, 'dsgcn_data_bn_fwd')
        ctx.save_for_backward(x, save)
        ctx.cfg = (int(mvc), bool(training), gamma is not None, beta is not None)
        ctx.defer_ok = _leafish(gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, save = ctx.saved_tensors
        mvc, training, has_g, has_b = ctx.cfg
        if not training:
            raise NotImplementedError('data_bn: backward through the eval-mode BatchNorm is not on this path')
        N, M, T, V, C = x.shape
        dy = _f32c(dy)
        part = torch.empty((2, N * M, V * C), device=x.device, dtype=torch.float32)
        rc = native.lib().dsgcn_data_bn_bwd(_ptr(x), _ptr(dy), _ptr(save[0]), _ptr(save[1]), _ptr(part[0]), _ptr(part[1]), N,
                                            M, T, V, C, mvc, _stream())
        native.check(rc, 'dsgcn_data_bn_bwd')
        rows = (N, M * V * C) if mvc else (N * M, V * C)
        dg = param_colsum(part[0].view(rows), ctx.defer_ok) if has_g else None
        db = param_colsum(part[1].view(rows), ctx.defer_ok) if has_b else None
        return None, dg, db, None, None, None


def data_bn_eligible(x, bn):
    """Can the input BatchNorm take the two-launch form?  (fp32 CUDA clip that needs no gradient, momentum-form buffers,
    V*C <= 256 channels per person, a (T, V*C) tile that fits LDS)"""
    return (FUSED_ENDS and isinstance(bn, torch.nn.BatchNorm1d) and x.is_cuda and x.dtype == torch.float32
            and not x.requires_grad and x.dim() == 5 and x.shape[3] * x.shape[4] <= 256
            and x.shape[2] * x.shape[3] * x.shape[4] * 4 <= 60 * 1024
            and (bn.momentum is not None or not bn.training or not bn.track_running_stats)
            and (bn.training or bn.running_mean is not None))


def data_bn(x, bn, bn_type):
    """x (N, M, T, V, C) -> (N*M, C, T, V): the backbones' input BatchNorm1d ('VC' | 'MVC') without the permute copies."""
    training = bn.training or not bn.track_running_stats or bn.running_mean is None
    return _DataBN.apply(x, bn.weight, bn.bias, bn, bn_type == 'MVC', training)
