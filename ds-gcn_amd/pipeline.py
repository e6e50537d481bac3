"""Skeleton input pipeline behind the reference's transform names and config dicts.

Reference: pyskl/datasets/pipelines/pose_related.py (``PreNormalize3D`` 250-336, ``PreNormalize2D`` 130-140,
``RandomRot`` 144-178, ``JointToBone``/``ToMotion``/``GenSkeFeat`` 340-442, ``PoseDecode`` 19-54, ``FormatGCNInput``
468-518), pipelines/sampling.py:10-192 (``UniformSample``), pipelines/formatting.py (``Collect``, ``ToTensor``),
pyskl/datasets/pose_dataset.py:89-125 (``PoseDataset`` pickle format).

Two ways to run the same pipeline config:

* ``Compose(cfg_list)(sample_dict)`` — one sample at a time on the host in numpy, the reference's data-loader-worker
  model.  Transform names, constructor kwargs, result-dict keys and the use of numpy's GLOBAL RNG (which draws, in
  which order) follow the reference, so a seeded run reproduces its frame indices and rotation angles.
* ``SkeletonBatcher(cfg_list)`` — the MI355X form.  At >= 3k clips/s per GPU an 8-worker numpy loader starves the
  node, so the raw clips stay RESIDENT in HBM (``SkeletonStore``: NTU-60 3D is 3.4 GB of the 288 GB), the host only makes
  the per-sample DECISIONS for a batch (valid frames / person order / body centre, rotation angles, frame indices: a few
  hundred bytes per clip, same RNG draws as above), and ONE HIP launch (``dsgcn_skeleton_prep``, csrc/skeleton.hip)
  does all per-element work — centre + mask, rotation, bone / motion features, frame gather, person padding, clip
  layout — writing the network input ``(N, clips, M, T, V, C)`` directly.  There is no CPU fallback for this form.

The HRNet-pose Kinetics-400 chain (configs/dsstgcn/kinetics400_hrnet/j.py:25-39: ``DecompressPose`` -> sample -> decode ->
``PoseCompact`` -> features) runs in both forms: a compressed pickle is unpacked ONCE when the store is built
(``decompress_detections``, vectorised), the per-frame joint extents are cached on the store, a batch's compact boxes are one
gather + min / max over the sampled frames (``compact_boxes``), and the shift rides in ``dsgcn_skeleton_prep``.

Not implemented (raise): 2-D heat-map transforms, ``float_ok`` sampling, memcached loading."""
import copy
import pickle

import numpy as np
import torch

from .registry import Registry

PIPELINES = Registry('pipeline')
DATASETS = Registry('dataset')

BONE_PAIRS = {
    'nturgb+d': ((0, 1), (1, 20), (2, 20), (3, 2), (4, 20), (5, 4), (6, 5), (7, 6), (8, 20), (9, 8), (10, 9), (11, 10),
                 (12, 0), (13, 12), (14, 13), (15, 14), (16, 0), (17, 16), (18, 17), (19, 18), (21, 22), (20, 20),
                 (22, 7), (23, 24), (24, 11)),
    'openpose': ((0, 0), (1, 0), (2, 1), (3, 2), (4, 3), (5, 1), (6, 5), (7, 6), (8, 2), (9, 8), (10, 9), (11, 5),
                 (12, 11), (13, 12), (14, 0), (15, 0), (16, 14), (17, 15)),
    'coco': ((0, 0), (1, 0), (2, 0), (3, 1), (4, 2), (5, 0), (6, 0), (7, 5), (8, 6), (9, 7), (10, 8), (11, 0), (12, 0),
             (13, 11), (14, 12), (15, 13), (16, 14)),
}


def bone_parent_table(dataset, V):
    """parent[v]: the joint subtracted from v to form its bone (identity where the layout defines none)."""
    if dataset not in BONE_PAIRS:
        raise ValueError(f'The dataset type {dataset} is not supported')
    parent = np.arange(V, dtype=np.int32)
    for v1, v2 in BONE_PAIRS[dataset]:
        parent[v1] = v2
    return parent


# ---------------------------------------------------------------------------------------------------------------
# per-sample decisions (shared by the host transforms and the batch planner)
# ---------------------------------------------------------------------------------------------------------------

def _axis_rotation(axis, theta):
    """Rotation about `axis` by `theta` (Euler-Rodrigues form); identity for a null axis or angle."""
    if np.abs(axis).sum() < 1e-6 or np.abs(theta) < 1e-6:
        return np.eye(3)
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.sqrt(np.dot(axis, axis))
    a = np.cos(theta / 2.0)
    b, c, d = -axis * np.sin(theta / 2.0)
    return np.array([[a * a + b * b - c * c - d * d, 2 * (b * c + a * d), 2 * (b * d - a * c)],
                     [2 * (b * c - a * d), a * a + c * c - b * b - d * d, 2 * (c * d + a * b)],
                     [2 * (b * d + a * c), 2 * (c * d - a * b), a * a + d * d - b * b - c * c]])


def _angle(v1, v2):
    if np.abs(v1).sum() < 1e-6 or np.abs(v2).sum() < 1e-6:
        return 0
    u1, u2 = v1 / np.linalg.norm(v1), v2 / np.linalg.norm(v2)
    return np.arccos(np.clip(np.dot(u1, u2), -1.0, 1.0))


def normalize3d_decision(kp, zaxis=(0, 1), xaxis=(8, 4), align_spine=True, align_center=True):
    """What PreNormalize3D decides for one raw clip ``kp (M, T, V, 3)``: -> dict(active, frames, swap, center, matrix).
    frames: indices of the kept (non-empty) frames of the leading person; swap: the two persons trade places (the second
    has more non-empty frames); center: the body centre subtracted from every non-zero joint; matrix: spine / shoulder
    alignment (identity when off).  active False: an all-zero-sum clip passes through untouched."""
    M, T, V, C = kp.shape
    if kp.sum() == 0:
        return dict(active=False, frames=np.arange(T), swap=False, center=np.zeros(3), matrix=np.eye(3))
    assert M in (1, 2)
    nonempty = ~np.all(np.isclose(kp, 0), axis=(2, 3))               # (M, T)
    idx0 = np.flatnonzero(nonempty[0])
    swap = False
    frames = idx0
    if M == 2:
        idx1 = np.flatnonzero(nonempty[1])
        if len(idx0) < len(idx1):
            frames, swap = idx1, True
    lead = kp[1 if swap else 0]
    first = lead[frames[0]]                                         # (V, 3) of the leading person's first kept frame
    center = np.zeros(3, dtype=kp.dtype)
    if align_center:
        center = first[1 if V == 25 else V - 1].copy()              # in the clip's own precision, like the reference
        first = (first - center) * ((first != 0).sum(-1) > 0)[:, None]
    first = first.astype(np.float64)
    matrix = np.eye(3)
    if align_spine:
        spine = first[zaxis[1]] - first[zaxis[0]]
        mz = _axis_rotation(np.cross(spine, [0, 0, 1]), _angle(spine, [0, 0, 1]))
        first = first @ mz.T
        shoulders = first[xaxis[0]] - first[xaxis[1]]
        mx = _axis_rotation(np.cross(shoulders, [1, 0, 0]), _angle(shoulders, [1, 0, 0]))
        matrix = mx @ mz
    return dict(active=True, frames=frames, swap=swap, center=center, matrix=matrix)


def euler_rotation(theta):
    """R = Rz Ry Rx for the three angles of RandomRot (pose_related.py:150-157)."""
    c, s = np.cos(theta), np.sin(theta)
    rx = np.array([[1, 0, 0], [0, c[0], s[0]], [0, -s[0], c[0]]])
    ry = np.array([[c[1], 0, -s[1]], [0, 1, 0], [s[1], 0, c[1]]])
    rz = np.array([[c[2], s[2], 0], [-s[2], c[2], 0], [0, 0, 1]])
    return rz @ ry @ rx


def euler_rotations(thetas):
    """``euler_rotation`` for a stack of angle triples (N, 3) -> (N, 3, 3) in one pass (batched matmul: the last bit of a
    float64 entry may differ from the 2-D product's, far below the fp32 the kernel receives)."""
    th = np.asarray(thetas, dtype=np.float64).reshape(-1, 3)
    c, s = np.cos(th), np.sin(th)
    n = len(th)
    rx = np.zeros((n, 3, 3)); ry = np.zeros((n, 3, 3)); rz = np.zeros((n, 3, 3))
    rx[:, 0, 0] = 1; rx[:, 1, 1] = c[:, 0]; rx[:, 1, 2] = s[:, 0]; rx[:, 2, 1] = -s[:, 0]; rx[:, 2, 2] = c[:, 0]
    ry[:, 1, 1] = 1; ry[:, 0, 0] = c[:, 1]; ry[:, 0, 2] = -s[:, 1]; ry[:, 2, 0] = s[:, 1]; ry[:, 2, 2] = c[:, 1]
    rz[:, 2, 2] = 1; rz[:, 0, 0] = c[:, 2]; rz[:, 0, 1] = s[:, 2]; rz[:, 1, 0] = -s[:, 2]; rz[:, 1, 1] = c[:, 2]
    return rz @ ry @ rx


def uniform_frame_indices(num_frames, clip_len, num_clips=1, test_mode=False, p_interval=(1, 1), seed=255):
    """Frame indices of UniformSampleFrames (sampling.py:49-151), consuming numpy's global RNG draw for draw like the
    reference: per clip one rand() (crop ratio), one randint (crop offset), then the case-specific draws.
    -> int array (num_clips * clip_len), already wrapped modulo num_frames."""
    if test_mode:
        np.random.seed(seed)
    out = []
    for clip in range(num_clips):
        full = num_frames
        ratio = np.random.rand() * (p_interval[1] - p_interval[0]) + p_interval[0]
        nf = int(ratio * full)
        off = np.random.randint(full - nf + 1)
        if nf < clip_len:
            if test_mode:
                start = clip if nf < num_clips else clip * nf // num_clips
            else:
                start = np.random.randint(0, nf)
            inds = np.arange(start, start + clip_len)
        elif nf < 2 * clip_len:
            picks = np.random.choice(clip_len + 1, nf - clip_len, replace=False)
            bump = np.zeros(clip_len + 1, dtype=np.int64)
            bump[picks] = 1
            inds = np.arange(clip_len) + np.cumsum(bump)[:-1]
        else:
            edges = np.arange(clip_len + 1) * nf // clip_len            # (= [i * nf // clip_len for i in ...])
            inds = edges[:clip_len] + np.random.randint(np.diff(edges))
        out.append(inds + off)
    return np.mod(np.concatenate(out), num_frames).astype(np.int64)


# ---------------------------------------------------------------------------------------------------------------
# host transforms (one sample, numpy)
# ---------------------------------------------------------------------------------------------------------------

@PIPELINES.register_module()
class Compose:

    def __init__(self, transforms):
        self.transforms = [PIPELINES.build(t) if isinstance(t, dict) else t for t in transforms]

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
        return data


@PIPELINES.register_module()
class PreNormalize3D:

    def __init__(self, zaxis=[0, 1], xaxis=[8, 4], align_spine=True, align_center=True):
        self.zaxis, self.xaxis, self.align_spine, self.align_center = zaxis, xaxis, align_spine, align_center

    def __call__(self, results):
        kp = results['keypoint']
        assert kp.shape[1] == results.get('total_frames', kp.shape[1])
        d = normalize3d_decision(kp, self.zaxis, self.xaxis, self.align_spine, self.align_center)
        if not d['active']:
            return results
        kp = kp[:, d['frames']]
        if d['swap']:
            kp = kp[[1, 0]]
        if self.align_center:
            kp = (kp - d['center']) * ((kp != 0).sum(-1) > 0)[..., None]
        if self.align_spine:
            kp = np.einsum('mtvc,kc->mtvk', kp, d['matrix'])
        results['keypoint'] = kp
        results['total_frames'] = kp.shape[1]
        results['body_center'] = d['center'].astype(results['keypoint'].dtype) if self.align_center else d['center']
        return results


@PIPELINES.register_module()
class PreNormalize2D:

    def __init__(self, img_shape=(1080, 1920)):
        self.img_shape = img_shape

    def __call__(self, results):
        h, w = results.get('img_shape', self.img_shape)
        kp = results['keypoint']
        kp[..., 0] = (kp[..., 0] - (w / 2)) / (w / 2)
        kp[..., 1] = (kp[..., 1] - (h / 2)) / (h / 2)
        return results


@PIPELINES.register_module()
class RandomRot:

    def __init__(self, theta=0.3):
        self.theta = theta

    def draw_angles(self, C):
        """The RNG draws of one call, as the reference makes them: three angles (C = 3) or one (C = 2) -> (3,) float64."""
        if C == 3:
            return np.random.uniform(-self.theta, self.theta, size=3)
        th = np.random.uniform(-self.theta)          # sic: the reference passes only `low` (high defaults to 1.0)
        return np.array([th, 0.0, 0.0])

    def draw(self, C):
        """The rotation matrix of one call (C x C), consuming the RNG as the reference does."""
        ang = self.draw_angles(C)
        if C == 3:
            return euler_rotation(ang)
        th = ang[0]
        return np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])

    def __call__(self, results):
        kp = results['keypoint']
        if np.all(np.isclose(kp, 0)):
            return results
        assert kp.shape[-1] in (2, 3)
        results['keypoint'] = np.einsum('ab,mtvb->mtva', self.draw(kp.shape[-1]), kp)
        return results


def skeleton_features(kp, dataset, feats):
    """Joint / bone / joint-motion / bone-motion features of ``kp (M, T, V, C)`` concatenated on the last axis
    (JointToBone, ToMotion, MergeSkeFeat: pose_related.py:340-412).  For 2-D layouts with a score channel the score of a
    bone / motion entry is the mean of the two scores it is built from."""
    M, T, V, C = kp.shape
    assert C in (2, 3)
    scored = C == 3 and dataset in ('openpose', 'coco')
    parts = {}
    if 'b' in feats or 'bm' in feats:
        parent = bone_parent_table(dataset, V)
        bone = np.zeros((M, T, V, C), dtype=np.float32)
        bone[...] = kp - kp[:, :, parent]
        if scored:
            bone[..., 2] = (kp[..., 2] + kp[:, :, parent, 2]) / 2
        parts['b'] = bone
    parts['j'] = kp

    def motion(x):
        m = np.zeros_like(x)
        m[:, :T - 1] = np.diff(x, axis=1)
        if scored:
            m[:, :T - 1, :, 2] = (x[:, :T - 1, :, 2] + x[:, 1:, :, 2]) / 2
        return m
    if 'jm' in feats:
        parts['jm'] = motion(parts['j'])
    if 'bm' in feats:
        parts['bm'] = motion(parts['b'])
    return np.concatenate([parts[f] for f in feats], axis=-1)


@PIPELINES.register_module()
class GenSkeFeat:

    def __init__(self, dataset='nturgb+d', feats=['j'], axis=-1):
        if axis != -1:
            raise NotImplementedError('GenSkeFeat: features are concatenated on the channel axis')
        bone_parent_table(dataset, 25 if dataset == 'nturgb+d' else (18 if dataset == 'openpose' else 17))
        self.dataset, self.feats, self.axis = dataset, list(feats), axis

    def __call__(self, results):
        if 'keypoint_score' in results and 'keypoint' in results:
            assert self.dataset != 'nturgb+d'
            assert results['keypoint'].shape[-1] == 2, 'Only 2D keypoints have keypoint_score. '
            results['keypoint'] = np.concatenate([results.pop('keypoint'), results.pop('keypoint_score')[..., None]], -1)
        results['keypoint'] = skeleton_features(results['keypoint'], self.dataset, self.feats)
        return results


def decompress_detections(det, frame_of, total_frames, squeeze=True, max_person=10):
    """Detection rows -> person-major arrays, for one clip, without a Python loop over detections.

    ``det (D, V, 3)`` = (x, y, score) per detection, ``frame_of (D,)`` non-decreasing frame ids.  -> ``(xy (M, T, V, 2) fp16,
    score (M, T, V) fp16, T, capped)``: detection d lands in person slot ``d - (first detection of its frame)``; with
    ``squeeze`` the frames that hold a detection are renumbered 0..T-1; M = the busiest frame's detection count, and when
    that exceeds ``max_person`` every frame's detections are re-ranked by their fp16 score total (high first, ties in
    detection order) and only the first ``max_person`` ranks stay.  Contract: pose_related.py:521-607."""
    fr = np.asarray(frame_of).astype(np.int64)
    D = len(fr)
    if D == 0:
        raise ValueError('DecompressPose: a clip without detections')
    if D > 1 and (fr[1:] < fr[:-1]).any():
        raise AssertionError('DecompressPose: detections must be ordered by frame (frame_inds non-decreasing)')
    if squeeze:
        fr = np.cumsum(np.concatenate([[0], fr[1:] != fr[:-1]]))          # rank of the frame among the occupied ones
        total_frames = int(fr[-1]) + 1
    total_frames = int(total_frames)
    slot = np.arange(D) - np.searchsorted(fr, fr, side='left')
    crowd, capped = int(slot.max()) + 1, False
    det = np.asarray(det)
    if crowd > max_person:
        # the reference ranks a frame's persons by the row sums of its fp16 score table: same dtype, same row length
        strength = np.ascontiguousarray(det[:, :, 2], dtype=np.float16).sum(-1)
        det = det[np.lexsort((-strength.astype(np.float32), fr))]          # stable: by frame, then score high -> low
        keep = slot < max_person                                          # (fr and slot are unchanged by that order)
        det, fr, slot, crowd, capped = det[keep], fr[keep], slot[keep], max_person, True
    V = det.shape[1]
    xy = np.zeros((crowd, total_frames, V, 2), dtype=np.float16)
    score = np.zeros((crowd, total_frames, V), dtype=np.float16)
    xy[slot, fr] = det[:, :, :2]
    score[slot, fr] = det[:, :, 2]
    return xy, score, total_frames, capped


@PIPELINES.register_module()
class DecompressPose:
    """The Kinetics pose pickles hold one row per DETECTION (``keypoint (D, V, 3)``, ``frame_inds (D,)``, optionally
    ``anno_inds`` = the detections a dataset's ``box_thr`` keeps); this unpacks a clip into ``keypoint (M, T, V, 2)`` +
    ``keypoint_score (M, T, V)`` (fp16).  See ``decompress_detections``; reference pose_related.py:521-607."""

    def __init__(self, squeeze=True, max_person=10):
        self.squeeze, self.max_person = squeeze, max_person

    def __call__(self, results):
        missing = [k for k in ('total_frames', 'frame_inds', 'keypoint') if k not in results]
        assert not missing, f'DecompressPose needs {missing}'
        det, frame_of = results['keypoint'], results.pop('frame_inds')
        chosen = results.get('anno_inds')
        if chosen is not None:
            det, frame_of = det[chosen], frame_of[chosen]
        xy, score, T, capped = decompress_detections(det, frame_of, results['total_frames'], self.squeeze, self.max_person)
        results.update(keypoint=xy, keypoint_score=score, total_frames=T)
        if capped:
            results['num_person'] = self.max_person
        return results


def nonzero_extent(xy, axes):
    """(…, 2) coordinates -> (lo_x, lo_y, hi_x, hi_y) over ``axes``, zeros not counted (+inf / -inf where nothing is)."""
    live = xy != 0
    lo = np.where(live, xy, np.inf).min(axis=axes)
    hi = np.where(live, xy, -np.inf).max(axis=axes)
    return np.concatenate([lo, hi], -1).astype(xy.dtype)


def compact_boxes(extent, img_hw, padding=0.25, threshold=10, hw_ratio=None, allow_imgpad=True, promotion='legacy'):
    """PoseCompact's box for a stack of clips at once.  ``extent (N, 4)`` = (lo_x, lo_y, hi_x, hi_y) of the non-zero joints
    in each clip's precision (fp32 after PoseDecode), ``img_hw (N, 2)``.  -> ``(apply (N,) bool, box (N, 4) int64 = x0, y0,
    x1, y1)``: the tight box grown by ``padding`` about its centre, stretched to ``hw_ratio``, truncated to integers (and to
    the image unless ``allow_imgpad``); ``apply`` False where the tight box is narrower than ``threshold`` either way.

    ``promotion``: the scalar code of augmentations.py:60-116 runs on numpy SCALARS, so its precision follows numpy's scalar
    promotion rule.  'legacy' (default) = NumPy 1.x — the only line the reference runs on (it uses ``np.Inf``, removed in
    2.0): centre and half extent are fp32, but ``fp32 scalar * (1 + padding)`` (a Python float) is float64, and so is
    everything after it.  'nep50' = NumPy >= 2: the Python floats adopt fp32 and the whole box stays fp32.  The two differ
    only when a box edge lies within an fp32 ulp of an integer (the ``int()`` truncation then lands a pixel apart)."""
    if promotion not in ('legacy', 'nep50'):
        raise ValueError("promotion must be 'legacy' or 'nep50'")
    e = np.asarray(extent)
    lo, hi = e[:, :2], e[:, 2:]
    with np.errstate(invalid='ignore'):
        span = hi - lo
        apply = ~((span[:, 0] < threshold) | (span[:, 1] < threshold))
        mid = (hi + lo) / 2
        half = span / 2
        if promotion == 'legacy':
            half = half.astype(np.float64)
        half = half * (1 + padding)
        if hw_ratio is not None:
            half_h = np.maximum(hw_ratio[0] * half[:, 0], half[:, 1])
            half_w = np.maximum(1 / hw_ratio[1] * half_h, half[:, 0])
            half = np.stack([half_w, half_h], 1)
        lo, hi = mid - half, mid + half
        if not allow_imgpad:
            wh = np.asarray(img_hw)[:, ::-1]
            lo, hi = np.maximum(0, lo), np.minimum(wh, hi)
        ok = apply[:, None]
        box = np.concatenate([np.trunc(np.where(ok, lo, 0)), np.trunc(np.where(ok, hi, 0))], 1).astype(np.int64)
    return apply, box


@PIPELINES.register_module()
class PoseCompact:
    """Move the coordinate origin to the corner of the box around all non-zero joints of the (sampled) clip — see
    ``compact_boxes`` — and record it: non-zero x / y are shifted, ``img_shape`` becomes the box size, ``crop_quadruple``
    composes with an earlier crop.  NaN coordinates become 0 first.  Reference: augmentations.py:21-116."""

    def __init__(self, padding=0.25, threshold=10, hw_ratio=None, allow_imgpad=True):
        assert padding >= 0
        if hw_ratio is not None and not isinstance(hw_ratio, (tuple, list)):
            hw_ratio = (hw_ratio, hw_ratio)
        self.padding, self.threshold, self.allow_imgpad = padding, threshold, allow_imgpad
        self.hw_ratio = None if hw_ratio is None else tuple(hw_ratio)

    def boxes(self, extent, img_hw):
        return compact_boxes(extent, img_hw, self.padding, self.threshold, self.hw_ratio, self.allow_imgpad)

    def __call__(self, results):
        kp = results['keypoint']
        np.nan_to_num(kp, copy=False, nan=0.0, posinf=np.inf, neginf=-np.inf)
        h, w = results['img_shape']
        apply, box = self.boxes(nonzero_extent(kp[..., :2], tuple(range(kp.ndim - 1)))[None], [(h, w)])
        if not apply[0]:
            return results
        x0, y0, x1, y1 = (int(b) for b in box[0])
        for axis, origin in ((0, x0), (1, y0)):
            col = kp[..., axis]
            col[col != 0] -= origin
        results['img_shape'] = (y1 - y0, x1 - x0)
        ox, oy, sw, sh = results.get('crop_quadruple', (0., 0., 1., 1.))
        results['crop_quadruple'] = (ox + sw * (x0 / w), oy + sh * (y0 / h), sw * ((x1 - x0) / w), sh * ((y1 - y0) / h))
        return results


@PIPELINES.register_module()
class UniformSampleFrames:

    def __init__(self, clip_len, num_clips=1, test_mode=False, float_ok=False, p_interval=1, seed=255):
        if float_ok:
            raise NotImplementedError('UniformSampleFrames(float_ok=True) is outside the skeleton configs')
        self.clip_len, self.num_clips, self.test_mode, self.seed = clip_len, num_clips, test_mode, seed
        self.float_ok = False
        self.p_interval = p_interval if isinstance(p_interval, tuple) else (p_interval, p_interval)

    def __call__(self, results):
        n = results['total_frames']
        if 'keypoint' in results:
            assert n == results['keypoint'].shape[1]
        inds = uniform_frame_indices(n, self.clip_len, self.num_clips, self.test_mode, self.p_interval, self.seed)
        results['frame_inds'] = (inds + results['start_index']).astype(np.int_)
        results['clip_len'] = self.clip_len
        results['frame_interval'] = None
        results['num_clips'] = self.num_clips
        return results


@PIPELINES.register_module()
class UniformSample(UniformSampleFrames):
    pass


@PIPELINES.register_module()
class PoseDecode:

    def __call__(self, results):
        if 'frame_inds' not in results:
            results['frame_inds'] = np.arange(results['total_frames'])
        inds = np.squeeze(results['frame_inds']) if results['frame_inds'].ndim != 1 else results['frame_inds']
        results['frame_inds'] = inds
        inds = inds + results.get('offset', 0)
        for key in ('keypoint_score', 'keypoint'):
            if key in results:
                results[key] = results[key][:, inds].astype(np.float32)
        return results


def format_persons(kp, num_person, mode):
    """(M, T, V, C) -> (num_person, T, V, C): zero (or first-person) padding, or truncation."""
    M = kp.shape[0]
    if M < num_person:
        kp = np.concatenate([kp, np.zeros((num_person - M,) + kp.shape[1:], dtype=kp.dtype)], 0)
        if mode == 'loop':
            kp[1:] = kp[0]
    elif M > num_person:
        kp = kp[:num_person]
    return kp


@PIPELINES.register_module()
class FormatGCNInput:

    def __init__(self, num_person=2, mode='zero'):
        assert mode in ['zero', 'loop']
        self.num_person, self.mode = num_person, mode

    def __call__(self, results):
        kp = results['keypoint']
        if 'keypoint_score' in results:
            kp = np.concatenate((kp, results['keypoint_score'][..., None]), axis=-1)
        kp = format_persons(kp, self.num_person, self.mode)
        M, T, V, C = kp.shape
        nc = results.get('num_clips', 1)
        assert T % nc == 0
        results['keypoint'] = np.ascontiguousarray(kp.reshape(M, nc, T // nc, V, C).transpose(1, 0, 2, 3, 4))
        return results


@PIPELINES.register_module()
class Collect:

    def __init__(self, keys, meta_keys=(), meta_name='img_metas', nested=False):
        self.keys, self.meta_keys, self.meta_name, self.nested = keys, meta_keys, meta_name, nested

    def __call__(self, results):
        data = {k: results[k] for k in self.keys}
        if len(self.meta_keys):
            data[self.meta_name] = {k: results[k] for k in self.meta_keys}
        if self.nested:
            data = {k: [v] for k, v in data.items()}
        return data


@PIPELINES.register_module()
class ToTensor:

    def __init__(self, keys):
        self.keys = keys

    def __call__(self, results):
        for k in self.keys:
            v = results[k]
            results[k] = v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v))
        return results


@DATASETS.register_module()
class PoseDataset(torch.utils.data.Dataset):
    """The reference's skeleton annotation pickle: ``{'split': {name: [ids]}, 'annotations': [{'frame_dir', 'label',
    'keypoint' (M, T, V, C), 'total_frames', ...}]}`` (or a bare annotation list).  ``dataset[i]`` runs the pipeline on a
    copy of sample i with ``start_index=0`` and ``modality='Pose'`` added, like BaseDataset.prepare_*_frames."""

    def __init__(self, ann_file, pipeline, split=None, valid_rate=1, valid_ratio=None, box_thr=0.5, class_prob=None,
                 memcached=False, mc_cfg=('localhost', 22077), test_mode=False, data_prefix='', **unsupported):
        """``valid_ratio`` / ``box_thr`` (the Kinetics pose pickles, pose_dataset.py:66-83,
        configs/dsstgcn/kinetics400_hrnet/j.py:22-23,80-82): a clip stays when ``valid[box_thr] / total_frames >=
        valid_ratio``; its ``anno_inds`` = ``box_score >= box_thr`` (what ``DecompressPose`` filters by); ``valid`` and
        ``box_score`` are dropped from every record.  ``memcached`` needs a memcached server holding the keypoints:
        rejected (load the pickle with its keypoints instead)."""
        if memcached:
            raise NotImplementedError('PoseDataset(memcached=True): no memcached client on this path — use an annotation '
                                      'file that carries the keypoints')
        if class_prob is not None:
            raise NotImplementedError('PoseDataset(class_prob=...) (class-balanced resampling) is not used by the '
                                      'skeleton configs')
        for key in list(unsupported):
            if key in ('num_classes', 'start_index', 'modality', 'multi_class', 'sample_by_class', 'power'):
                if unsupported[key] not in (None, False, 0, 'Pose'):
                    raise NotImplementedError(f'PoseDataset({key}={unsupported[key]!r}) is outside this path')
                unsupported.pop(key)
        if unsupported:
            raise TypeError(f'PoseDataset got unexpected arguments {sorted(unsupported)}')
        if box_thr is not None and box_thr not in (.5, .6, .7, .8, .9):
            raise AssertionError('box_thr must be one of 0.5, 0.6, 0.7, 0.8, 0.9 or None')
        self.ann_file, self.split, self.test_mode = ann_file, split, test_mode
        self.valid_rate, self.valid_ratio, self.box_thr, self.data_prefix = valid_rate, valid_ratio, box_thr, data_prefix
        self.pipeline = pipeline if callable(pipeline) else Compose(pipeline)
        self.start_index, self.modality = 0, 'Pose'
        with open(ann_file, 'rb') as f:
            data = pickle.load(f)
        if split:
            names, data = data['split'][split], data['annotations']
            if valid_rate:
                names = names[0:int(len(names) * valid_rate)]
            names = set(names)
            key = 'filename' if 'filename' in data[0] else 'frame_dir'
            data = [x for x in data if x[key] in names]
        if data_prefix:
            import os.path as osp
            for item in data:
                for key in ('filename', 'frame_dir'):
                    if key in item:
                        item[key] = osp.join(data_prefix, item[key])
        if valid_ratio is not None:
            assert isinstance(valid_ratio, float)
            data = [x for x in data if x['valid'][box_thr] / x['total_frames'] >= valid_ratio]
            for item in data:
                item['anno_inds'] = item['box_score'] >= box_thr
        for item in data:
            item.pop('valid', None)
            item.pop('box_score', None)
        self.video_infos = data

    def __len__(self):
        return len(self.video_infos)

    def sample_info(self, idx):
        info = copy.deepcopy(self.video_infos[idx])
        info['modality'], info['start_index'] = self.modality, self.start_index
        info['test_mode'] = self.test_mode
        return info

    def __getitem__(self, idx):
        return self.pipeline(self.sample_info(idx))


# ---------------------------------------------------------------------------------------------------------------
# MI355X form: clips resident in HBM, host decisions, one HIP launch per batch
# ---------------------------------------------------------------------------------------------------------------

class SkeletonStore:
    """Raw clips of a dataset packed into ONE device buffer (ragged: each clip keeps its own M and T).
    ``annotations``: sequence of dicts with 'keypoint' (M, T, V, C) [+ 'label', 'total_frames'] and, for 2-D pose pickles,
    'keypoint_score' (M, T, V) and 'img_shape' (h, w).  A score rides as channel 2 behind the two coordinates (what
    ``GenSkeFeat`` builds, pose_related.py:427-432); ``coordC`` is the number of coordinate channels the geometric
    transforms act on, ``C`` the channels per joint in the buffer.

    Compressed Kinetics annotations (``keypoint (D, V, 3)`` per DETECTION + ``frame_inds`` [+ ``anno_inds``]) are unpacked
    here, once per clip, by ``decompress_detections(**decompress)`` (default: ``DecompressPose``'s own defaults); the
    parameters are remembered in ``self.decompressed`` and a batcher whose pipeline asks for other ones refuses the store.
    ``plan_only``: host tables without the device buffer (``SkeletonBatcher.plan`` works, ``run`` does not)."""

    def __init__(self, annotations, device='cuda', decompress=None, plan_only=False):
        self.decompressed = None
        if len(annotations) and np.asarray(annotations[0]['keypoint']).ndim == 3:
            opt = dict(squeeze=True, max_person=10)
            opt.update(decompress or {})
            self.decompressed = (bool(opt['squeeze']), int(opt['max_person']))
            unpacked = []
            for a in annotations:
                det, frame_of = np.asarray(a['keypoint']), np.asarray(a['frame_inds'])
                if a.get('anno_inds') is not None:
                    det, frame_of = det[a['anno_inds']], frame_of[a['anno_inds']]
                xy, score, _, _ = decompress_detections(det, frame_of, a['total_frames'], *self.decompressed)
                unpacked.append(dict(a, keypoint=xy, keypoint_score=score))
            annotations = unpacked
        elif decompress is not None:
            raise ValueError('SkeletonStore(decompress=...): the annotations are not in the per-detection layout')
        self.src = [np.ascontiguousarray(a['keypoint']) for a in annotations]          # original dtype (see normalize2d)
        self.scores = [a.get('keypoint_score') for a in annotations]
        self.img_shapes = [tuple(a['img_shape']) if a.get('img_shape') is not None else None for a in annotations]
        self.V, self.coordC = self.src[0].shape[2], self.src[0].shape[3]
        assert all(k.shape[2:] == (self.V, self.coordC) for k in self.src), 'clips must share the joint layout'
        has_score = [s is not None for s in self.scores]
        if any(has_score):
            if not all(has_score) or self.coordC != 2:
                raise ValueError("'keypoint_score' must accompany every clip, and only 2-D keypoints carry one")
        self.C = self.coordC + (1 if any(has_score) else 0)
        self.M = np.array([k.shape[0] for k in self.src], dtype=np.int32)
        self.T = np.array([k.shape[1] for k in self.src], dtype=np.int32)
        sizes = np.array([k.size // self.coordC * self.C for k in self.src], dtype=np.int64)
        self.offset = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        self.labels = np.array([int(a.get('label', -1)) for a in annotations], dtype=np.int64)
        self.device = device
        self.norm2d_shape = None
        self.plan_only = bool(plan_only)
        self._extent = None
        if not plan_only and not torch.device(device).type == 'cuda':
            raise RuntimeError('SkeletonStore keeps the clips in HBM: it needs a CUDA/ROCm device (no CPU fallback)')
        self._upload(self.src)

    def _upload(self, kps):
        host = []
        for k, sc in zip(kps, self.scores):
            k32 = k.astype(np.float32)
            if sc is not None:
                k32 = np.concatenate([k32, np.asarray(sc, dtype=np.float32)[..., None]], -1)
            host.append(np.ascontiguousarray(k32))
        self.host = host                      # the decisions read a few frames of the clip on the host
        self._extent = None
        if self.plan_only:
            self.data = None
            return
        self.data = torch.from_numpy(np.concatenate([k.reshape(-1) for k in host])).to(self.device)
        if not self.data.is_cuda:
            raise RuntimeError('SkeletonStore keeps the clips in HBM: it needs a CUDA/ROCm device (no CPU fallback)')

    def normalize2d(self, default_img_shape):
        """``PreNormalize2D`` (pose_related.py:130-140) for the whole store, once: the transform is deterministic, so it is
        applied when the pipeline first asks for it instead of per epoch — on the host, in each clip's OWN dtype and with
        its own ``img_shape`` (the reference normalises in place: an fp16 pickle is rounded to fp16 here too), then the
        buffer is re-uploaded."""
        shape = tuple(default_img_shape)
        if self.norm2d_shape is not None:
            if self.norm2d_shape != shape:
                raise ValueError('SkeletonStore was already normalised with another default img_shape')
            return
        if self.coordC != 2:
            raise ValueError('PreNormalize2D applies to 2-D keypoints')
        out = []
        for k, ishape in zip(self.src, self.img_shapes):
            h, w = ishape if ishape is not None else shape
            k = k.copy()
            k[..., 0] = (k[..., 0] - (w / 2)) / (w / 2)
            k[..., 1] = (k[..., 1] - (h / 2)) / (h / 2)
            out.append(k)
        self.norm2d_shape = shape
        self._upload(out)

    def frame_extents(self):
        """What ``PoseCompact`` needs of a clip, made once: NaN coordinates zeroed in the resident copy (the transform does
        that to every sample it sees) and, per frame, (lo_x, lo_y, hi_x, hi_y) of the non-zero coordinates over all persons
        and joints.  -> (rows (sum T, 4) float32, first_row (n,) int64): the box of ANY set of sampled frames is a gather +
        min / max over these rows."""
        if self._extent is None:
            if self.coordC != 2:
                raise ValueError('PoseCompact applies to 2-D keypoints')
            if any(np.isnan(k[..., :2]).any() for k in self.host):
                self.src = [np.where(np.isnan(k), np.zeros((), k.dtype), k) for k in self.src]
                self._upload(self.src)
            rows = [nonzero_extent(k[..., :2], (0, 2)) for k in self.host]
            first = np.concatenate([[0], np.cumsum(self.T.astype(np.int64))[:-1]])
            self._extent = (np.concatenate(rows).astype(np.float32), first)
        return self._extent

    def __len__(self):
        return len(self.src)


class SkeletonBatcher:
    """Runs a reference-style skeleton pipeline config for a whole batch with one HIP launch.

    Two transform orders are understood (the last three transforms of either only fix the output layout):
    * the 3-D / pre-extracted 2-D configs: [PreNormalize3D | PreNormalize2D] -> [RandomRot] -> GenSkeFeat ->
      UniformSample(Frames) -> PoseDecode -> FormatGCNInput -> Collect -> ToTensor;
    * the compressed Kinetics config: DecompressPose -> UniformSampleFrames -> PoseDecode -> [PoseCompact] -> GenSkeFeat ->
      FormatGCNInput -> Collect -> ToTensor — the features are built from the SAMPLED frames here (a motion feature
      differences neighbours of the sampled sequence) in fp32 (PoseDecode has cast the clip).
    ``plan(store, indices)`` makes the per-clip decisions on the host (same numpy RNG draws, in the same order, as running
    ``Compose`` clip by clip); ``run(store, plan)`` launches the kernel."""

    _GEOMETRY = {'PreNormalize3D': 0, 'PreNormalize2D': 0, 'RandomRot': 1, 'GenSkeFeat': 2, 'UniformSample': 3,
                 'UniformSampleFrames': 3}
    _SAMPLED = {'DecompressPose': 0, 'UniformSample': 1, 'UniformSampleFrames': 1, 'PoseDecode': 2, 'PoseCompact': 3,
                'GenSkeFeat': 4}

    def __init__(self, pipeline):
        self.norm3d = self.norm2d = self.rot = self.feat = self.sample = self.decompress = self.compact = None
        self.num_person, self.person_mode = 2, 'zero'
        order = []
        for cfg in pipeline:
            typ = cfg['type']
            kw = {k: v for k, v in cfg.items() if k != 'type'}
            order.append(typ)
            if typ == 'PreNormalize3D':
                self.norm3d = PreNormalize3D(**kw)
            elif typ == 'PreNormalize2D':
                self.norm2d = PreNormalize2D(**kw)
            elif typ == 'RandomRot':
                self.rot = RandomRot(**kw)
            elif typ == 'GenSkeFeat':
                self.feat = GenSkeFeat(**kw)
            elif typ in ('UniformSample', 'UniformSampleFrames'):
                self.sample = UniformSampleFrames(**kw)
            elif typ == 'DecompressPose':
                self.decompress = DecompressPose(**kw)
            elif typ == 'PoseCompact':
                self.compact = PoseCompact(**kw)
            elif typ == 'FormatGCNInput':
                self.num_person, self.person_mode = kw.get('num_person', 2), kw.get('mode', 'zero')
            elif typ not in ('PoseDecode', 'Collect', 'ToTensor'):
                raise NotImplementedError(f'SkeletonBatcher: transform {typ} has no batched HIP form')
        if self.feat is None or self.sample is None:
            raise ValueError('SkeletonBatcher needs GenSkeFeat and UniformSample in the pipeline')
        # features of the sampled sequence (Kinetics order) or of the source clip (every other shipped config)?
        self.sampled_first = order.index('GenSkeFeat') > min(order.index(t) for t in order if t.startswith('UniformSample'))
        rank = self._SAMPLED if self.sampled_first else self._GEOMETRY
        seen = [t for t in order if t not in ('FormatGCNInput', 'Collect', 'ToTensor') and not
                (t == 'PoseDecode' and not self.sampled_first)]
        if any(t not in rank for t in seen) or [rank[t] for t in seen] != sorted(rank[t] for t in seen):
            raise NotImplementedError(
                'SkeletonBatcher: transforms must come as normalise, rotate, features, sample — or, for compressed pose '
                'pickles, decompress, sample, decode, compact, features')
        if self.sampled_first and 'PoseDecode' not in order:
            raise NotImplementedError('SkeletonBatcher: features of the sampled frames need PoseDecode in front of them')

    # ---- host decisions ---------------------------------------------------------------------------------------------
    # A clip's geometry decisions (kept frames, person order, centre, alignment matrix, "nothing to rotate") depend on the
    # clip alone: they are made once per clip and kept on the store; what a batch costs on the host afterwards is the RNG
    # draws — clip by clip, in the reference's order — and a few array gathers over the whole batch.  (Round 3 redid the
    # per-clip numpy work every epoch: 3.4 k clips/s on one thread, below the 5 k clips/s step; profiles/r04/pipeline.txt.)

    def _table(self, store):
        n3 = self.norm3d
        key = ('none', self.rot is not None) if n3 is None else ('norm3d', tuple(n3.zaxis), tuple(n3.xaxis),
                                                                  bool(n3.align_spine), bool(n3.align_center))
        cache = store.__dict__.setdefault('_decisions', {})
        if key not in cache:
            n, tmax = len(store.host), int(max(k.shape[1] for k in store.host))
            if n3 is None:
                tmax = 0                                  # every frame is kept: no per-clip frame list to remember
            cache[key] = dict(done=np.zeros(n, bool), frames=np.zeros((n, tmax), np.int32), nf=np.zeros(n, np.int32),
                              swap=np.zeros(n, bool), masked=np.zeros(n, bool), allzero=np.zeros(n, bool),
                              center=np.zeros((n, 3)), matrix=np.tile(np.eye(3), (n, 1, 1)),
                              f16=np.array([k.dtype == np.float16 for k in store.src]))
        return cache[key]

    def _decide(self, store, tab, idx):
        """The deterministic decisions of clip ``idx`` (what round 3's plan() recomputed per batch)."""
        kp = store.host[idx][..., :store.coordC]          # coordinates only: a score channel is not geometry
        T = kp.shape[1]
        frames, swap, masked = np.arange(T), False, False
        center, mat = np.zeros(3), np.eye(3)
        allzero = self.rot is not None and bool(np.all(np.isclose(kp, 0)))       # only RandomRot asks
        if self.norm3d is not None:
            n = self.norm3d
            d = normalize3d_decision(kp, n.zaxis, n.xaxis, n.align_spine, n.align_center)
            if d['active']:
                frames, swap = d['frames'], d['swap']
                if n.align_center:
                    center, masked = d['center'], True
                    kept = kp[:, frames]            # RandomRot's "nothing to rotate" test sees the centred clip
                    allzero = bool(np.all(np.isclose((kept - center) * ((kept != 0).sum(-1) > 0)[..., None], 0)))
                mat = d['matrix']
        if self.norm3d is not None:
            tab['frames'][idx, :len(frames)] = frames
        tab['nf'][idx], tab['swap'][idx], tab['masked'][idx], tab['allzero'][idx] = len(frames), swap, masked, allzero
        tab['center'][idx], tab['matrix'][idx] = center, mat
        tab['done'][idx] = True

    def prepare(self, store, indices=None):
        """Make the per-clip decisions ahead of time (all clips, or ``indices``): the first epoch then runs at the rate of
        the later ones."""
        if self.norm2d is not None:
            store.normalize2d(self.norm2d.img_shape)      # once per store (deterministic): per-clip img_shape, source dtype
        if self.norm3d is not None and store.coordC != 3:
            raise ValueError('PreNormalize3D needs 3-D keypoints')
        want = None if self.decompress is None else (bool(self.decompress.squeeze), int(self.decompress.max_person))
        if want != getattr(store, 'decompressed', None):
            raise ValueError(f'SkeletonBatcher: the pipeline asks for DecompressPose{want}, the store was built with '
                             f'{getattr(store, "decompressed", None)} (SkeletonStore(annotations, decompress=dict(...)))')
        if self.compact is not None:
            store.frame_extents()                         # once per store: NaN -> 0, per-frame joint extents
            if not self.compact.allow_imgpad and any(s is None for s in store.img_shapes):
                raise ValueError('PoseCompact(allow_imgpad=False) needs every clip\'s img_shape')
        tab = self._table(store)
        todo = np.arange(len(tab['done'])) if indices is None else np.unique(np.asarray(indices, dtype=np.int64))
        for i in todo[~tab['done'][todo]]:
            self._decide(store, tab, int(i))
        return tab

    def plan(self, store, indices):
        """Host decisions for the clips ``indices`` of ``store`` -> dict of small numpy arrays (see dsgcn_skeleton_prep).
        numpy's global RNG is consumed exactly as running the reference's ``Compose`` clip by clip would: per clip the
        rotation angles (unless there is nothing to rotate), then the sampler's draws."""
        idx = np.asarray(indices, dtype=np.int64)
        N = len(idx)
        clip_len, num_clips = self.sample.clip_len, self.sample.num_clips
        F = num_clips * clip_len
        tab = self.prepare(store, idx)
        nf = tab['nf'][idx]
        C = store.coordC
        rotated = np.zeros(N, bool)
        if self.rot is not None:
            rotated = ~tab['allzero'][idx]
        thetas = np.zeros((N, 3))
        inds = np.empty((N, F), np.int64)
        smp = self.sample
        for row in range(N):                              # the RNG stream: nothing else happens clip by clip
            if rotated[row]:
                thetas[row] = self.rot.draw_angles(C)
            inds[row] = uniform_frame_indices(int(nf[row]), clip_len, num_clips, smp.test_mode, smp.p_interval, smp.seed)
        mat = tab['matrix'][idx]
        if rotated.any():
            if C == 3:
                rot = euler_rotations(thetas)
            else:
                c, s_ = np.cos(thetas[:, 0]), np.sin(thetas[:, 0])
                rot = np.tile(np.eye(3), (N, 1, 1))
                rot[:, 0, 0], rot[:, 0, 1], rot[:, 1, 0], rot[:, 1, 1] = c, -s_, s_, c
            mat = np.where(rotated[:, None, None], rot @ mat, mat)
        if self.norm3d is not None:
            frames = tab['frames'][idx]                   # (N, Tmax), valid up to nf
            f0 = np.take_along_axis(frames, inds, 1)
        else:
            frames, f0 = None, inds
        if self.sampled_first:
            # the difference partner of a sampled frame is the NEXT SAMPLED frame (clips still concatenated: ToMotion runs
            # in front of FormatGCNInput's split), the last one has none
            f1 = np.concatenate([f0[:, 1:], np.full((N, 1), -1, f0.dtype)], 1)
        else:
            nxt = inds + 1
            has_next = nxt < nf[:, None]
            nxt = np.where(has_next, nxt, 0)
            f1 = np.where(has_next, nxt if frames is None else np.take_along_axis(frames, nxt, 1), -1)
        half = tab['f16'][idx] & ~rotated & (self.norm3d is None) & (not self.sampled_first)      # fp16 feature arithmetic
        flags = tab['swap'][idx].astype(np.int32) | (tab['masked'][idx].astype(np.int32) << 1) | (half.astype(np.int32) << 2)
        center = tab['center'][idx].astype(np.float32)
        extra = {}
        if self.compact is not None:
            # the box of the SAMPLED frames: per-frame extents gathered and reduced for the whole batch, integer origin
            # subtracted from every non-zero coordinate by the kernel (flag bit 3)
            rows, first = store.frame_extents()
            ext = rows[first[idx][:, None] + f0]          # (N, F, 4)
            extent = np.concatenate([ext[..., :2].min(1), ext[..., 2:].max(1)], 1)
            hw = np.array([s if s is not None else (0, 0) for s in (store.img_shapes[i] for i in idx)], dtype=np.int64)
            apply, box = self.compact.boxes(extent, hw)
            if (np.abs(box) >= 1 << 24).any():
                raise ValueError('PoseCompact: box origin beyond fp32 integer range')
            center = np.zeros((N, 3), np.float32)
            center[:, :2] = box[:, :2]
            flags = flags | (apply.astype(np.int32) << 3)
            extra = dict(box=box, compacted=apply)
        return dict(offset=store.offset[idx].astype(np.int64), M=store.M[idx].astype(np.int32), T=store.T[idx].astype(np.int32),
                    flags=flags, center=center, matrix=mat.reshape(N, 9).astype(np.float32),
                    f0=f0.astype(np.int32), f1=f1.astype(np.int32), label=store.labels[idx].astype(np.int64), **extra)

    def run(self, store, plan):
        """-> (keypoint (N, clips, num_person, clip_len, V, C_out) float32 on the store's device, label (N, 1) int64)."""
        from . import native
        if store.data is None:
            raise RuntimeError('SkeletonStore(plan_only=True) holds no device buffer: nothing to run the kernel on')
        dev = store.data.device
        N, F = plan['f0'].shape
        feats = self.feat.feats
        code = {'j': 0, 'b': 1, 'jm': 2, 'bm': 3}
        C = store.C
        out = torch.empty((N, self.sample.num_clips, self.num_person, self.sample.clip_len, store.V, C * len(feats)),
                          device=dev, dtype=torch.float32)
        parent = torch.from_numpy(bone_parent_table(self.feat.dataset, store.V)).to(dev)
        scored = int(C == 3 and self.feat.dataset in ('openpose', 'coco'))
        # one small H2D copy of the packed decisions (a few hundred bytes per clip)
        ints = np.concatenate([plan['M'], plan['T'], plan['flags'], plan['f0'].reshape(-1), plan['f1'].reshape(-1)]).astype(np.int32)
        flts = np.concatenate([plan['center'].reshape(-1), plan['matrix'].reshape(-1)]).astype(np.float32)
        d_int = torch.from_numpy(ints).to(dev, non_blocking=True)
        d_flt = torch.from_numpy(flts).to(dev, non_blocking=True)
        d_off = torch.from_numpy(plan['offset']).to(dev, non_blocking=True)
        fcodes = (code[f] for f in feats)
        fmask = 0
        for i, c in enumerate(fcodes):
            fmask |= c << (2 * i)
        st = torch.cuda.current_stream().cuda_stream
        ip = d_int.data_ptr()
        rc = native.lib().dsgcn_skeleton_prep(
            store.data.data_ptr(), d_off.data_ptr(), ip, ip + 4 * N, ip + 8 * N, ip + 12 * N, ip + 12 * N + 4 * N * F,
            d_flt.data_ptr(), d_flt.data_ptr() + 12 * N, parent.data_ptr(), out.data_ptr(), N, self.sample.num_clips,
            self.num_person, self.sample.clip_len, store.V, C, len(feats), fmask, scored,
            1 if self.person_mode == 'loop' else 0, st)
        native.check(rc, 'dsgcn_skeleton_prep')
        label = torch.from_numpy(plan['label']).to(dev, non_blocking=True).view(N, 1)
        return out, label

    def __call__(self, store, indices):
        return self.run(store, self.plan(store, indices))


def build_dataset(cfg):
    return DATASETS.build(cfg)
