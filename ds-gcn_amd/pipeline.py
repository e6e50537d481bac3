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

``DecompressPose`` (pose_related.py:521-607) and ``PoseCompact`` (augmentations.py:21-116) — the two extra transforms of the
HRNet-pose Kinetics-400 config (configs/dsstgcn/kinetics400_hrnet/j.py:25-39) — exist in the host (``Compose``) form only.

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


@PIPELINES.register_module()
class DecompressPose:
    """Kinetics pose pickles store one row per DETECTION: ``keypoint (D, V, 3)`` = (x, y, score) with ``frame_inds (D,)``
    saying which frame a detection belongs to (and ``anno_inds``, the detections a dataset's ``box_thr`` keeps).  ->
    ``keypoint (M, T, V, 2)`` / ``keypoint_score (M, T, V)`` in fp16 with the persons of a frame in detection order, M =
    the largest number of detections in one frame, capped at ``max_person`` by total score.  ``squeeze``: frames without
    a detection are dropped (frame indices renumbered densely).  Reference: pose_related.py:521-607."""

    def __init__(self, squeeze=True, max_person=10):
        self.squeeze, self.max_person = squeeze, max_person

    def __call__(self, results):
        for k in ('total_frames', 'frame_inds', 'keypoint'):
            assert k in results
        total_frames = results['total_frames']
        frame_inds = results.pop('frame_inds')
        keypoint = results['keypoint']
        if 'anno_inds' in results:
            frame_inds = frame_inds[results['anno_inds']]
            keypoint = keypoint[results['anno_inds']]
        assert np.all(np.diff(frame_inds) >= 0), 'frame_inds should be monotonical increasing'
        if self.squeeze:
            uni = np.unique(frame_inds)
            frame_inds = np.searchsorted(uni, frame_inds).astype(np.int16)
            total_frames = np.max(frame_inds) + 1
        results['total_frames'] = total_frames
        num_joints = keypoint.shape[1]
        num_person = int(np.bincount(np.asarray(frame_inds, dtype=np.int64)).max())      # = scipy.stats.mode(...).count
        new_kp = np.zeros([num_person, total_frames, num_joints, 2], dtype=np.float16)
        new_kpscore = np.zeros([num_person, total_frames, num_joints], dtype=np.float16)
        nperson_per_frame = np.zeros([total_frames], dtype=np.int16)
        for frame_ind, kp in zip(frame_inds, keypoint):
            person_ind = nperson_per_frame[frame_ind]
            new_kp[person_ind, frame_ind] = kp[:, :2]
            new_kpscore[person_ind, frame_ind] = kp[:, 2]
            nperson_per_frame[frame_ind] += 1
        if num_person > self.max_person:
            for i in range(total_frames):
                nperson = nperson_per_frame[i]
                score_sum = new_kpscore[:nperson, i].sum(-1)
                inds = sorted(range(nperson), key=lambda x: -score_sum[x])
                new_kpscore[:nperson, i] = new_kpscore[inds, i]
                new_kp[:nperson, i] = new_kp[inds, i]
            num_person = self.max_person
            results['num_person'] = num_person
        results['keypoint'] = new_kp[:num_person]
        results['keypoint_score'] = new_kpscore[:num_person]
        return results


@PIPELINES.register_module()
class PoseCompact:
    """Crop the coordinate frame to the tight box around all non-zero joints of the (sampled) clip, grown by ``padding``
    and to the ``hw_ratio`` asked for: joints are shifted by the box origin, ``img_shape`` becomes the box size,
    ``crop_quadruple`` records it.  Boxes narrower than ``threshold`` pixels leave the sample untouched.  Reference:
    augmentations.py:21-116."""

    def __init__(self, padding=0.25, threshold=10, hw_ratio=None, allow_imgpad=True):
        self.padding, self.threshold, self.allow_imgpad = padding, threshold, allow_imgpad
        if hw_ratio is not None and not isinstance(hw_ratio, (tuple, list)):
            hw_ratio = (hw_ratio, hw_ratio)
        self.hw_ratio = tuple(hw_ratio) if hw_ratio is not None else None
        assert self.padding >= 0

    def __call__(self, results):
        h, w = results['img_shape']
        kp = results['keypoint']
        kp[np.isnan(kp)] = 0.
        kp_x, kp_y = kp[..., 0], kp[..., 1]
        min_x = np.min(kp_x[kp_x != 0], initial=np.inf)
        min_y = np.min(kp_y[kp_y != 0], initial=np.inf)
        max_x = np.max(kp_x[kp_x != 0], initial=-np.inf)
        max_y = np.max(kp_y[kp_y != 0], initial=-np.inf)
        if max_x - min_x < self.threshold or max_y - min_y < self.threshold:
            return results
        center = ((max_x + min_x) / 2, (max_y + min_y) / 2)
        half_width = (max_x - min_x) / 2 * (1 + self.padding)
        half_height = (max_y - min_y) / 2 * (1 + self.padding)
        if self.hw_ratio is not None:
            half_height = max(self.hw_ratio[0] * half_width, half_height)
            half_width = max(1 / self.hw_ratio[1] * half_height, half_width)
        min_x, max_x = center[0] - half_width, center[0] + half_width
        min_y, max_y = center[1] - half_height, center[1] + half_height
        if not self.allow_imgpad:
            min_x, min_y = int(max(0, min_x)), int(max(0, min_y))
            max_x, max_y = int(min(w, max_x)), int(min(h, max_y))
        else:
            min_x, min_y = int(min_x), int(min_y)
            max_x, max_y = int(max_x), int(max_y)
        kp_x[kp_x != 0] -= min_x
        kp_y[kp_y != 0] -= min_y
        results['img_shape'] = (max_y - min_y, max_x - min_x)
        a = results.get('crop_quadruple', (0., 0., 1., 1.))
        b = (min_x / w, min_y / h, (max_x - min_x) / w, (max_y - min_y) / h)
        results['crop_quadruple'] = (a[0] + a[2] * b[0], a[1] + a[3] * b[1], a[2] * b[2], a[3] * b[3])
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
    transforms act on, ``C`` the channels per joint in the buffer."""

    def __init__(self, annotations, device='cuda'):
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
        if not torch.device(device).type == 'cuda':
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

    def __len__(self):
        return len(self.src)


class SkeletonBatcher:
    """Runs a reference-style skeleton pipeline config for a whole batch with one HIP launch.

    Understood transforms, in the reference's order: [PreNormalize3D | PreNormalize2D] -> [RandomRot] -> GenSkeFeat ->
    UniformSample(Frames) -> PoseDecode -> FormatGCNInput -> Collect -> ToTensor (the last three only fix the output
    layout).  ``plan(store, indices)`` makes the per-clip decisions on the host (same numpy RNG draws, in the same order,
    as running ``Compose`` clip by clip); ``run(store, plan)`` launches the kernel."""

    def __init__(self, pipeline):
        self.norm3d = self.norm2d = self.rot = self.feat = self.sample = None
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
            elif typ == 'FormatGCNInput':
                self.num_person, self.person_mode = kw.get('num_person', 2), kw.get('mode', 'zero')
            elif typ not in ('PoseDecode', 'Collect', 'ToTensor'):
                raise NotImplementedError(f'SkeletonBatcher: transform {typ} has no batched HIP form')
        if self.feat is None or self.sample is None:
            raise ValueError('SkeletonBatcher needs GenSkeFeat and UniformSample in the pipeline')
        want = [t for t in order if t in ('PreNormalize3D', 'PreNormalize2D', 'RandomRot', 'GenSkeFeat', 'UniformSample',
                                          'UniformSampleFrames')]
        rank = {'PreNormalize3D': 0, 'PreNormalize2D': 0, 'RandomRot': 1, 'GenSkeFeat': 2, 'UniformSample': 3,
                'UniformSampleFrames': 3}
        if [rank[t] for t in want] != sorted(rank[t] for t in want):
            raise NotImplementedError('SkeletonBatcher: transforms must come in the order normalise, rotate, features, sample')

    # ---- host decisions ---------------------------------------------------------------------------------------------
    # A clip's geometry decisions (kept frames, person order, centre, alignment matrix, "nothing to rotate") depend on the
    # clip alone: they are made once per clip and kept on the store; what a batch costs on the host afterwards is the RNG
    # draws — clip by clip, in the reference's order — and a few array gathers over the whole batch.  (Round 3 redid the
    # per-clip numpy work every epoch: 3.4 k clips/s on one thread, below the 5 k clips/s step; profiles/r04/pipeline.txt.)

    def _table(self, store):
        n3 = self.norm3d
        key = ('none',) if n3 is None else ('norm3d', tuple(n3.zaxis), tuple(n3.xaxis), bool(n3.align_spine),
                                            bool(n3.align_center))
        cache = store.__dict__.setdefault('_decisions', {})
        if key not in cache:
            n, tmax = len(store.host), int(max(k.shape[1] for k in store.host))
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
        allzero = bool(np.all(np.isclose(kp, 0)))
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
        frames = tab['frames'][idx]                       # (N, Tmax), valid up to nf
        nxt = inds + 1
        has_next = nxt < nf[:, None]
        f0 = np.take_along_axis(frames, inds, 1)
        f1 = np.where(has_next, np.take_along_axis(frames, np.where(has_next, nxt, 0), 1), -1)
        half = tab['f16'][idx] & ~rotated & (self.norm3d is None)      # fp16 feature arithmetic
        flags = tab['swap'][idx].astype(np.int32) | (tab['masked'][idx].astype(np.int32) << 1) | (half.astype(np.int32) << 2)
        return dict(offset=store.offset[idx].astype(np.int64), M=store.M[idx].astype(np.int32), T=store.T[idx].astype(np.int32),
                    flags=flags, center=tab['center'][idx].astype(np.float32), matrix=mat.reshape(N, 9).astype(np.float32),
                    f0=f0.astype(np.int32), f1=f1.astype(np.int32), label=store.labels[idx].astype(np.int64))

    def run(self, store, plan):
        """-> (keypoint (N, clips, num_person, clip_len, V, C_out) float32 on the store's device, label (N, 1) int64)."""
        from . import native
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
