"""Synthetic raw clips and pipeline configs of the input-pipeline fixtures (tests/golden/pipeline.npz): shared by the
generator (reference side) and the tests, so only the reference's OUTPUTS are stored."""
import numpy as np

CLIP_LEN = 16


def raw_clips():
    """Seven (M, T, 25, 3) float32 clips covering PreNormalize3D's and UniformSample's branches."""
    rng = np.random.RandomState(2024)

    def body(M, T):
        base = rng.randn(1, 1, 25, 3).astype(np.float32) * 0.4 + np.array([0.2, 0.1, 3.0], dtype=np.float32)
        walk = np.cumsum(rng.randn(M, T, 1, 3).astype(np.float32) * 0.02, axis=1)
        return (base + walk + rng.randn(M, T, 25, 3).astype(np.float32) * 0.05).astype(np.float32)
    clips = []
    a = body(2, 80)                                  # two full persons, T >= 2 * clip_len (block sampling)
    clips.append(a)
    b = body(2, 43)
    b[1] = 0                                         # second person absent (the common NTU case)
    b[0, :5] = 0                                     # leading empty frames are dropped
    clips.append(b)
    c = body(2, 30)
    c[0, 10:] = 0                                    # person 1 has more valid frames: the persons swap
    clips.append(c)
    clips.append(body(1, 11))                        # shorter than the clip: looped indices
    d = body(2, 50)
    d[0, 7, 3] = 0                                   # isolated zero joints stay zero after centring (mask)
    d[1, 20:25] = 0
    clips.append(d)
    clips.append(body(1, 25))                        # clip_len <= T < 2 * clip_len
    clips.append(np.zeros((2, 20, 25, 3), dtype=np.float32))     # all-zero clip passes through
    return clips


def annotations():
    return [dict(frame_dir=f'clip{i}', label=(7 * i) % 60, keypoint=k, total_frames=k.shape[1]) for i, k in enumerate(raw_clips())]


def pipelines():
    def pipe(norm=None, rot=0.2, feats=('j',), clips=1, test_mode=False, fmt=None):
        p = [dict(type='PreNormalize3D', **(norm if norm is not None else dict(align_spine=False)))]
        if rot:
            p.append(dict(type='RandomRot', theta=rot))
        p += [dict(type='GenSkeFeat', feats=list(feats)),
              dict(type='UniformSample', clip_len=CLIP_LEN, num_clips=clips, test_mode=test_mode),
              dict(type='PoseDecode'), dict(type='FormatGCNInput', **(fmt or {})),
              dict(type='Collect', keys=['keypoint', 'label'], meta_keys=[]), dict(type='ToTensor', keys=['keypoint'])]
        return p
    return {
        'train_j': pipe(), 'train_b': pipe(feats=('b',)), 'train_jm': pipe(feats=('jm',)), 'train_bm': pipe(feats=('bm',)),
        'train_all': pipe(feats=('j', 'b', 'jm', 'bm')),
        'val_j': pipe(rot=0), 'test10_j': pipe(rot=0, clips=10), 'test10_seeded': pipe(rot=0, clips=10, test_mode=True),
        'spine_j': pipe(norm=dict()), 'loop_j': pipe(fmt=dict(num_person=2, mode='loop')),
        'three_persons': pipe(fmt=dict(num_person=3)),
        'loop_three': pipe(fmt=dict(num_person=3, mode='loop')),      # M = 2 < 3 slots: the real second person is overwritten
    }


# ---- 2-D pose clips (HRNet / coco layout): keypoint (M, T, 17, 2) + keypoint_score (M, T, 17) + img_shape ----------------

def annotations_2d():
    """Five coco clips in pixel coordinates: fp16 and fp32 storage (PreNormalize2D works in place, in the pickle's dtype),
    with and without a per-clip img_shape, one or two persons."""
    rng = np.random.RandomState(77)
    out = []
    for i, (M, T, dt, shape) in enumerate([(2, 40, np.float16, (480, 854)), (1, 21, np.float32, None),
                                           (2, 16, np.float16, None), (1, 9, np.float32, (720, 1280)),
                                           (2, 33, np.float32, (1080, 1920))]):
        h, w = shape or (1080, 1920)
        base = np.array([w * 0.5, h * 0.5]) + rng.randn(1, 1, 17, 2) * np.array([w * 0.08, h * 0.15])
        kp = (base + np.cumsum(rng.randn(M, T, 1, 2) * 3.0, axis=1) + rng.randn(M, T, 17, 2) * 2.0).astype(dt)
        score = (rng.rand(M, T, 17) * 0.6 + 0.4).astype(dt)
        a = dict(frame_dir=f'pose{i}', label=(11 * i) % 400, keypoint=kp, keypoint_score=score, total_frames=T)
        if shape is not None:
            a['img_shape'] = shape
        out.append(a)
    return out


def pipelines_2d():
    def pipe(rot=0.0, feats=('j',), clips=1, shape=(1080, 1920)):
        p = [dict(type='PreNormalize2D', img_shape=shape)]
        if rot:
            p.append(dict(type='RandomRot', theta=rot))
        p += [dict(type='GenSkeFeat', dataset='coco', feats=list(feats)),
              dict(type='UniformSample', clip_len=CLIP_LEN, num_clips=clips), dict(type='PoseDecode'),
              dict(type='FormatGCNInput', num_person=2),
              dict(type='Collect', keys=['keypoint', 'label'], meta_keys=[]), dict(type='ToTensor', keys=['keypoint'])]
        return p
    return {'coco_j': pipe(), 'coco_all': pipe(feats=('j', 'b', 'jm', 'bm')), 'coco_rot_b': pipe(rot=0.2, feats=('b',)),
            'coco_test3': pipe(clips=3)}


# ---- compressed Kinetics pose annotations (HRNet detections: one row per detection) --------------------------------------

def annotations_k400():
    """Four clips in the compressed layout of data/k400/k400_hrnet.pkl: keypoint (D, 17, 3) fp16 = (x, y, score) per
    detection, frame_inds (D,), box_score (D,), valid {thr: frames with a box above thr}; some frames without any
    detection (squeeze), one clip with more persons in a frame than DecompressPose(max_person=2) keeps."""
    rng = np.random.RandomState(91)
    out = []
    for i, (T, maxp) in enumerate([(30, 2), (22, 1), (40, 3), (18, 2)]):
        frames, kps, bs = [], [], []
        for t in range(T):
            if (t + i) % 7 == 3:
                continue                                        # a frame without detections
            for _ in range(int(rng.randint(1, maxp + 1))):
                xy = rng.rand(17, 2) * np.array([180, 120]) + np.array([60 + 3 * i, 40])
                sc = rng.rand(17, 1) * 0.6 + 0.3
                k = np.concatenate([xy, sc], 1)
                if rng.rand() < 0.1:
                    k[rng.randint(17)] = 0                      # a missing joint
                frames.append(t); kps.append(k); bs.append(rng.rand() * 0.6 + 0.4)
        bs = np.array(bs, dtype=np.float32)
        fi = np.array(frames, dtype=np.int16)
        valid = {thr: int(len(np.unique(fi[bs >= thr]))) for thr in (0.5, 0.6, 0.7, 0.8, 0.9)}
        out.append(dict(frame_dir=f'k{i}', label=(37 * i) % 400, img_shape=(240, 320), original_shape=(240, 320),
                        total_frames=T, frame_inds=fi, keypoint=np.stack(kps).astype(np.float16), box_score=bs, valid=valid))
    return out


def pipelines_k400():
    """The transform chains of configs/dsstgcn/kinetics400_hrnet/j.py:25-72 (clip_len shortened)."""
    def pipe(clips=1, max_person=10, **compact):
        return [dict(type='DecompressPose', squeeze=True, max_person=max_person),
                dict(type='UniformSampleFrames', clip_len=CLIP_LEN, num_clips=clips), dict(type='PoseDecode'),
                dict(type='PoseCompact', **(compact or dict(hw_ratio=1., allow_imgpad=True))),
                dict(type='GenSkeFeat', dataset='coco', feats=['j']), dict(type='FormatGCNInput', num_person=2),
                dict(type='Collect', keys=['keypoint', 'label'], meta_keys=[]), dict(type='ToTensor', keys=['keypoint'])]
    return {'k400_train': pipe(), 'k400_test3': pipe(clips=3), 'k400_cap2': pipe(max_person=2),
            'k400_nopad': pipe(padding=0.1, threshold=10, hw_ratio=(0.8, 1.2), allow_imgpad=False)}
