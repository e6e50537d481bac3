"""Input pipeline against outputs of the REFERENCE's transforms (tests/golden/pipeline.npz, written by
tests/golden/gen_golden_pipeline.py from the imported pyskl/datasets/pipelines): the host (numpy, per sample) form on CPU,
the batched HIP form (-m gpu) on the same configs with the same RNG seeds."""
import copy
import os
import pickle
import sys

import numpy as np
import pytest
import torch

import dsgcn_amd as D
from dsgcn_amd import pipeline as P

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
sys.path.insert(0, GOLD)
from pipeline_cases import CLIP_LEN, annotations, annotations_2d, pipelines, pipelines_2d, raw_clips  # noqa: E402

Z = dict(np.load(os.path.join(GOLD, 'pipeline.npz')))
NAMES = list(pipelines())
NAMES_2D = list(pipelines_2d())


@pytest.mark.parametrize('name', NAMES)
def test_host_pipeline_vs_reference(name):
    """numpy transforms, one sample at a time, numpy's global RNG seeded like the generator: frame choice, rotation
    angles, person order and every value agree with the reference (float data to 1e-6: the reference rotates in fp64
    and rounds once, evaluation order inside einsum may differ in the last bit)."""
    pipe = P.Compose(copy.deepcopy(pipelines()[name]))
    np.random.seed(1000 + NAMES.index(name))
    for si, ann in enumerate(annotations()):
        sample = copy.deepcopy(ann)
        sample.update(start_index=0, modality='Pose')
        res = pipe(sample)
        want = Z[f'{name}_{si}']
        got = res['keypoint']
        assert torch.is_tensor(got) and got.dtype == torch.float32 and tuple(got.shape) == want.shape
        assert res['label'] == ann['label']
        assert np.abs(got.numpy() - want).max() < 1e-6, (name, si)


@pytest.mark.parametrize('name', NAMES_2D)
def test_host_pipeline_2d_vs_reference(name):
    """2-D pose pickles (coco layout): PreNormalize2D with the per-clip img_shape in the pickle's own dtype (fp16 clips
    keep their fp16 rounding, like the reference's in-place arithmetic), the score channel appended by GenSkeFeat."""
    pipe = P.Compose(copy.deepcopy(pipelines_2d()[name]))
    np.random.seed(2000 + NAMES_2D.index(name))
    for si, ann in enumerate(annotations_2d()):
        sample = copy.deepcopy(ann)
        sample.update(start_index=0, modality='Pose')
        res = pipe(sample)
        want = Z[f'{name}_{si}']
        assert tuple(res['keypoint'].shape) == want.shape and want.shape[-1] % 3 == 0      # x, y, score per feature
        assert np.abs(res['keypoint'].numpy() - want).max() < 1e-6, (name, si)


def test_frame_indices_branches_and_rng_stream():
    """UniformSample draws: every branch (T < clip, clip <= T < 2 clip, T >= 2 clip), several clips, seeded test mode —
    indices in range, sorted within a clip where the reference's are, and the test-mode stream is reproducible."""
    np.random.seed(3)
    for T in (5, 16, 20, 31, 32, 100):
        inds = P.uniform_frame_indices(T, 16, 3)
        assert inds.shape == (48,) and inds.min() >= 0 and inds.max() < T
        if T >= 16:
            assert all((np.diff(inds[c * 16:(c + 1) * 16]) >= 0).all() for c in range(3))
    a = P.uniform_frame_indices(57, 16, 10, test_mode=True)
    np.random.seed(99)
    b = P.uniform_frame_indices(57, 16, 10, test_mode=True)
    assert np.array_equal(a, b)


def test_pose_dataset_pickle_and_split(tmp_path):
    """PoseDataset reads the reference's annotation pickle ({'split': ..., 'annotations': [...]}) and applies the split."""
    anns = annotations()
    path = tmp_path / 'toy.pkl'
    with open(path, 'wb') as f:
        pickle.dump(dict(split=dict(train=[a['frame_dir'] for a in anns[:5]], val=[a['frame_dir'] for a in anns[5:]]),
                         annotations=anns), f)
    ds = D.build_dataset(dict(type='PoseDataset', ann_file=str(path), pipeline=pipelines()['val_j'], split='train'))
    assert len(ds) == 5 and len(D.PoseDataset(str(path), pipelines()['val_j'], split='val')) == 2
    np.random.seed(1000 + NAMES.index('val_j'))
    item = ds[0]
    assert np.abs(item['keypoint'].numpy() - Z['val_j_0']).max() < 1e-6 and item['label'] == anns[0]['label']
    with pytest.raises(NotImplementedError):
        D.PoseDataset(str(path), pipelines()['val_j'], split='val', memcached=True)


def test_batcher_rejects_what_it_cannot_batch():
    with pytest.raises(NotImplementedError):
        P.SkeletonBatcher([dict(type='PreNormalize3D'), dict(type='RandomScale', scale=0.1), dict(type='GenSkeFeat'),
                           dict(type='UniformSample', clip_len=8)])
    with pytest.raises(NotImplementedError):
        P.SkeletonBatcher([dict(type='GenSkeFeat'), dict(type='PreNormalize3D'), dict(type='UniformSample', clip_len=8)])
    with pytest.raises(RuntimeError):
        P.SkeletonStore(annotations(), device='cpu')            # the clips live in HBM: no CPU form


@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
def test_batched_hip_pipeline_vs_reference(name):
    """One dsgcn_skeleton_prep launch for the whole batch == the reference's per-sample transforms (same seed, same RNG
    draws on the host): fp32 on the device vs the reference's fp64-then-round: 2e-6 absolute on O(1) coordinates."""
    store = P.SkeletonStore(annotations())
    batcher = P.SkeletonBatcher(pipelines()[name])
    np.random.seed(1000 + NAMES.index(name))
    kp, label = batcher(store, list(range(len(store))))
    kp = kp.cpu().numpy()
    assert label.shape == (len(store), 1) and label[:, 0].tolist() == [a['label'] for a in annotations()]
    for si in range(len(store)):
        want = Z[f'{name}_{si}']
        assert kp[si].shape == want.shape
        assert np.abs(kp[si] - want).max() < 2e-6, (name, si, np.abs(kp[si] - want).max())


@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES_2D)
def test_batched_hip_pipeline_2d_vs_reference(name):
    """the batched form on 2-D pose pickles — keypoint_score carried as channel 2, PreNormalize2D with each
    clip's own img_shape, fp16 clips' features in fp16 arithmetic — against the reference's per-sample transforms."""
    anns = annotations_2d()
    store = P.SkeletonStore(anns)
    assert store.C == 3 and store.coordC == 2
    batcher = P.SkeletonBatcher(pipelines_2d()[name])
    np.random.seed(2000 + NAMES_2D.index(name))
    kp, label = batcher(store, list(range(len(store))))
    kp = kp.cpu().numpy()
    assert label[:, 0].tolist() == [a['label'] for a in anns]
    for si in range(len(store)):
        want = Z[f'{name}_{si}']
        assert kp[si].shape == want.shape
        assert np.abs(kp[si] - want).max() < 2e-6, (name, si, np.abs(kp[si] - want).max())


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['k400_train', 'k400_test3', 'k400_cap2', 'k400_nopad'])
def test_batched_hip_pipeline_k400_vs_reference(name):
    """BASELINE config 5's data section in the resident form: compressed detections unpacked into HBM once,
    compact boxes from cached per-frame extents, the shift inside dsgcn_skeleton_prep — bit-exact against the reference's
    per-sample DecompressPose -> UniformSampleFrames -> PoseDecode -> PoseCompact -> GenSkeFeat -> FormatGCNInput."""
    from pipeline_cases import pipelines_k400
    names = list(pipelines_k400())
    store = _k400_store(name)
    batcher = P.SkeletonBatcher(pipelines_k400()[name])
    np.random.seed(3000 + names.index(name))
    kp, label = batcher(store, list(range(len(store))))
    kp = kp.cpu().numpy()
    for si in range(len(store)):
        want = Z[f'{name}_{si}']
        assert kp[si].shape == want.shape
        assert np.array_equal(kp[si], want), (name, si, np.abs(kp[si] - want).max())


@pytest.mark.gpu
def test_batched_hip_pipeline_k400_motion_features_follow_the_sampled_frames():
    """With the features built AFTER sampling (the Kinetics order), jm / bm difference neighbours of the SAMPLED sequence:
    the launch against the host Compose form of the same chain (itself pinned to the reference by the k400 fixtures for j and
    by skeleton_features' fixtures for b / jm / bm)."""
    from pipeline_cases import pipelines_k400
    pipe = copy.deepcopy(pipelines_k400()['k400_test3'])
    for t in pipe:
        if t['type'] == 'GenSkeFeat':
            t['feats'] = ['j', 'b', 'jm', 'bm']
    store = _k400_store('k400_test3')
    np.random.seed(5)
    kp, _ = P.SkeletonBatcher(pipe)(store, [3, 0, 2])
    np.random.seed(5)
    host = P.Compose(copy.deepcopy(pipe))
    anns = annotations_k400()
    for row, si in enumerate([3, 0, 2]):
        sample = copy.deepcopy(anns[si])
        sample['anno_inds'] = sample.pop('box_score') >= 0.5
        sample.pop('valid')
        sample.update(start_index=0, modality='Pose')
        want = host(sample)['keypoint'].numpy()
        assert np.array_equal(kp[row].cpu().numpy(), want), (si, np.abs(kp[row].cpu().numpy() - want).max())


@pytest.mark.gpu
def test_batched_pipeline_feeds_the_recognizer():
    """(N, clips, M, T, V, C) straight from the HIP pipeline into RecognizerGCN.train_step / forward_test."""
    from bench import ds_cfg
    store = P.SkeletonStore(annotations())
    np.random.seed(0)
    torch.manual_seed(0)
    cfg = ds_cfg(60)
    cfg['backbone'].update(base_channels=16, num_stages=4, inflate_stages=[3], down_stages=[3])
    cfg['cls_head']['in_channels'] = 32
    m = D.build_model(cfg).cuda().train()
    kp, label = P.SkeletonBatcher(pipelines()['train_j'])(store, [0, 1, 2, 4])
    out = m.train_step(dict(keypoint=kp, label=label), None)
    out['loss'].backward()
    assert np.isfinite(out['log_vars']['loss'])
    m.eval()
    kp10, _ = P.SkeletonBatcher(pipelines()['test10_j'])(store, [0, 5])
    probs = m(keypoint=kp10, return_loss=False)
    assert probs.shape == (2, 60) and np.allclose(probs.sum(1), 1, atol=1e-5)


class _HostStore:
    """What SkeletonBatcher.plan() reads of a SkeletonStore, without the device buffer (the plan is host arithmetic)."""

    def __init__(self, anns):
        self.src = [np.ascontiguousarray(a['keypoint']) for a in anns]
        self.host = [k.astype(np.float32) for k in self.src]
        self.V, self.coordC = self.src[0].shape[2], self.src[0].shape[3]
        self.C = self.coordC
        self.M = np.array([k.shape[0] for k in self.src], dtype=np.int32)
        self.T = np.array([k.shape[1] for k in self.src], dtype=np.int32)
        sizes = np.array([k.size for k in self.src], dtype=np.int64)
        self.offset = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        self.labels = np.array([int(a.get('label', -1)) for a in anns], dtype=np.int64)

    def __len__(self):
        return len(self.src)


def _plan_clip_by_clip(b, store, indices):
    """The decisions made one clip at a time, transform by transform, the way the reference's Compose visits a sample
    (PreNormalize3D -> RandomRot -> UniformSample): the yardstick for the batched plan()."""
    rows = []
    for idx in indices:
        kp = store.host[idx][..., :store.coordC]
        T = kp.shape[1]
        frames, swap, masked, center, mat = np.arange(T), False, False, np.zeros(3), np.eye(3)
        allzero = bool(np.all(np.isclose(kp, 0)))
        if b.norm3d is not None:
            n = b.norm3d
            d = P.normalize3d_decision(kp, n.zaxis, n.xaxis, n.align_spine, n.align_center)
            if d['active']:
                frames, swap = d['frames'], d['swap']
                if n.align_center:
                    center, masked = d['center'], True
                    kept = kp[:, frames]
                    allzero = bool(np.all(np.isclose((kept - center) * ((kept != 0).sum(-1) > 0)[..., None], 0)))
                mat = d['matrix']
        rotated = b.rot is not None and not allzero
        if rotated:
            mat = b.rot.draw(3) @ mat
        inds = P.uniform_frame_indices(len(frames), b.sample.clip_len, b.sample.num_clips, b.sample.test_mode,
                                       b.sample.p_interval, b.sample.seed)
        nxt = np.where(inds + 1 < len(frames), inds + 1, -1)
        rows.append(dict(f0=frames[inds], f1=np.where(nxt >= 0, frames[np.maximum(nxt, 0)], -1), mat=mat, center=center,
                         flags=(1 if swap else 0) | (2 if masked else 0)))
    return rows


@pytest.mark.parametrize('name', NAMES)
def test_batched_plan_matches_clip_by_clip_decisions(name):
    """plan() makes the per-clip geometry decisions once (cached on the store) and the rest for the whole batch at once; the
    RNG stream, every frame index and flag must equal the clip-by-clip walk bit for bit — on a first visit, on a revisit
    from the cache, and for a permuted batch — and the matrices to the last fp32 bit but one."""
    store = _HostStore(annotations())
    b = P.SkeletonBatcher(pipelines()[name])
    for order in (list(range(len(store))), list(range(len(store)))[::-1], [2, 0, 2, 1]):
        np.random.seed(77)
        got = b.plan(store, order)
        after = np.random.rand()                         # the stream position after the batch
        np.random.seed(77)
        want = _plan_clip_by_clip(b, store, order)
        assert np.random.rand() == after
        for r, w in enumerate(want):
            assert np.array_equal(got['f0'][r], w['f0']) and np.array_equal(got['f1'][r], w['f1'])
            assert int(got['flags'][r]) & 3 == w['flags']
            assert np.allclose(got['matrix'][r], w['mat'].reshape(-1).astype(np.float32), rtol=0, atol=2e-7)
            assert np.array_equal(got['center'][r], np.asarray(w['center'], dtype=np.float32))
        assert got['label'].tolist() == [int(store.labels[i]) for i in order]
        assert got['offset'].tolist() == [int(store.offset[i]) for i in order]


def test_pose_dataset_valid_ratio_and_box_thr(tmp_path):
    """The Kinetics pose pickles' options (reference pose_dataset.py:66-83, configs/dsstgcn/kinetics400_hrnet/j.py:22-23,80-82):
    a clip stays when valid[box_thr] / total_frames >= valid_ratio, gets anno_inds = box_score >= box_thr, and loses its
    'valid' / 'box_score' bookkeeping; without valid_ratio nothing is filtered (the val / test splits of that config)."""
    rng = np.random.RandomState(0)
    anns = []
    for i, (valid5, total) in enumerate([(8, 10), (5, 10), (10, 10), (0, 12)]):
        anns.append(dict(frame_dir=f'v{i}', label=i, total_frames=total, img_shape=(240, 320), original_shape=(240, 320),
                         keypoint=rng.rand(6, 17, 3).astype(np.float16), frame_inds=np.array([0, 0, 1, 2, 3, 3], dtype=np.int16),
                         box_score=np.array([.9, .55, .45, .7, .2, .5], dtype=np.float32),
                         valid={0.5: valid5, 0.6: max(valid5 - 2, 0), 0.7: 1, 0.8: 0, 0.9: 0}))
    path = tmp_path / 'k.pkl'
    with open(path, 'wb') as f:
        pickle.dump(anns, f)
    ident = lambda r: r       # noqa: E731
    ds = D.PoseDataset(str(path), ident, valid_ratio=0.6, box_thr=0.5)
    assert [a['frame_dir'] for a in ds.video_infos] == ['v0', 'v2']                     # 8/10 and 10/10 pass, 5/10 and 0/12 do not
    for a in ds.video_infos:
        assert a['anno_inds'].tolist() == [True, True, False, True, False, True]
        assert 'valid' not in a and 'box_score' not in a
    item = ds[1]
    assert item['label'] == 2 and item['modality'] == 'Pose' and item['start_index'] == 0 and 'anno_inds' in item
    ds6 = D.PoseDataset(str(path), ident, valid_ratio=0.3, box_thr=0.6)
    assert [a['frame_dir'] for a in ds6.video_infos] == ['v0', 'v1', 'v2'] and ds6.video_infos[0]['anno_inds'].sum() == 2
    with open(path, 'wb') as f:
        pickle.dump(anns, f)
    plain = D.PoseDataset(str(path), ident, box_thr=0.5)                                # val / test: box_thr given, no valid_ratio
    assert len(plain) == 4 and all('anno_inds' not in a and 'valid' not in a for a in plain.video_infos)
    with pytest.raises(AssertionError):
        D.PoseDataset(str(path), ident, box_thr=0.55)
    with pytest.raises(NotImplementedError):
        D.PoseDataset(str(path), ident, memcached=True, mc_cfg=('localhost', 11211))


from pipeline_cases import annotations_k400, pipelines_k400  # noqa: E402
NAMES_K400 = list(pipelines_k400())


@pytest.mark.parametrize('name', NAMES_K400)
def test_host_pipeline_k400_vs_reference(name):
    """The HRNet-pose Kinetics-400 chain of BASELINE config 5's data section (configs/dsstgcn/kinetics400_hrnet/j.py:25-72):
    DecompressPose (detections -> (M, T, V) persons, frames without a detection squeezed out, the person cap) ->
    UniformSampleFrames -> PoseDecode -> PoseCompact (tight box, padding, hw_ratio, with and without image padding) ->
    GenSkeFeat(coco) -> FormatGCNInput, against the reference's own transforms on the same seeded RNG stream."""
    pipe = P.Compose(copy.deepcopy(pipelines_k400()[name]))
    np.random.seed(3000 + NAMES_K400.index(name))
    for si, ann in enumerate(annotations_k400()):
        sample = copy.deepcopy(ann)
        sample['anno_inds'] = sample.pop('box_score') >= 0.5
        sample.pop('valid')
        sample.update(start_index=0, modality='Pose')
        got = pipe(sample)['keypoint'].numpy()
        want = Z[f'{name}_{si}']
        assert got.shape == want.shape and got.dtype == want.dtype
        assert np.array_equal(got, want), (name, si, np.abs(got - want).max())      # fp16 storage, integer crops: bit-exact


def _k400_store(name, **kw):
    """The compressed clips the way PoseDataset(box_thr=0.5, valid_ratio=0.0) hands them over (anno_inds set), unpacked into a
    resident store with the DecompressPose parameters of pipeline `name`."""
    anns = []
    for ann in annotations_k400():
        a = copy.deepcopy(ann)
        a['anno_inds'] = a.pop('box_score') >= 0.5
        a.pop('valid')
        anns.append(a)
    dec = {k: v for k, v in pipelines_k400()[name][0].items() if k != 'type'}
    return P.SkeletonStore(anns, decompress=dec, **kw)


@pytest.mark.parametrize('name', NAMES_K400)
def test_batched_plan_k400_vs_reference(name):
    """The resident / batched form of config 5's chain, host half: the store unpacks the detections once, plan() draws the
    sampler's RNG clip by clip and gets every clip's compact box from the cached per-frame extents.  A numpy statement of
    what the launch does with those decisions (gather f0, shift the non-zero coordinates by the box origin, pad / cut the
    persons, split the clips) must reproduce the reference's transforms bit for bit — and leave the RNG where they do."""
    store = _k400_store(name, plan_only=True)
    b = P.SkeletonBatcher(pipelines_k400()[name])
    assert b.sampled_first and store.C == 3 and store.coordC == 2
    np.random.seed(3000 + NAMES_K400.index(name))
    plan = b.plan(store, list(range(len(store))))
    after = np.random.rand()
    np.random.seed(3000 + NAMES_K400.index(name))
    host = P.Compose(copy.deepcopy(pipelines_k400()[name]))
    for si, ann in enumerate(annotations_k400()):
        sample = copy.deepcopy(ann)
        sample['anno_inds'] = sample.pop('box_score') >= 0.5
        sample.pop('valid')
        sample.update(start_index=0, modality='Pose')
        res = host(sample)
    assert np.random.rand() == after
    nc = b.sample.num_clips
    for si in range(len(store)):
        kp = store.host[si][:, plan['f0'][si]].copy()                       # (M, F, V, 3)
        assert np.array_equal(plan['f1'][si][:-1], plan['f0'][si][1:]) and plan['f1'][si][-1] == -1
        if plan['flags'][si] & 8:
            for c in (0, 1):
                col = kp[..., c]
                col[col != 0] -= plan['center'][si][c]
        else:
            assert not plan['center'][si].any()
        kp = P.format_persons(kp, 2, 'zero')
        M, F, V, C = kp.shape
        got = kp.reshape(M, nc, F // nc, V, C).transpose(1, 0, 2, 3, 4)
        assert np.array_equal(got, Z[f'{name}_{si}']), (name, si)
    assert plan['compacted'].all()
    with pytest.raises(ValueError):                                          # other DecompressPose parameters than the store's
        P.SkeletonBatcher([dict(pipelines_k400()[name][0], max_person=7)] + pipelines_k400()[name][1:]).plan(store, [0])
    with pytest.raises(RuntimeError):
        b.run(store, plan)                                                   # plan_only: no device buffer


def test_decompress_and_compact_edge_cases():
    """What the fixtures do not reach: score ties under the person cap keep detection order; squeeze=False keeps the frame
    axis; boxes under the threshold leave the clip alone; an all-zero clip has no box; out-of-order detections raise."""
    det = np.zeros((5, 17, 3), np.float16)
    det[:, :, 2] = np.array([.5, .75, .5, .75, .25], np.float16)[:, None]
    det[:, :, 0] = np.arange(1, 6, dtype=np.float16)[:, None]
    fr = np.array([2, 2, 2, 2, 5], np.int16)
    xy, sc, T, capped = P.decompress_detections(det, fr, 9, squeeze=True, max_person=3)
    assert (T, capped, xy.shape) == (2, True, (3, 2, 17, 2))
    assert xy[:, 0, 0, 0].tolist() == [2., 4., 1.] and xy[:, 1, 0, 0].tolist() == [5., 0., 0.]      # .75 .75 then the first .5
    xy, sc, T, capped = P.decompress_detections(det, fr, 9, squeeze=False, max_person=10)
    assert (T, capped, xy.shape) == (9, False, (4, 9, 17, 2)) and xy[:, 2, 0, 0].tolist() == [1., 2., 3., 4.]
    with pytest.raises(AssertionError):
        P.decompress_detections(det, fr[::-1], 9)
    pc = P.PoseCompact(hw_ratio=1.)
    ext = np.array([[10, 10, 15, 200], [np.inf, np.inf, -np.inf, -np.inf], [10, 20, 110, 60]], np.float32)
    apply, box = pc.boxes(ext, [(240, 320)] * 3)
    assert apply.tolist() == [False, False, True] and not box[:2].any()
    assert box[2].tolist() == [-2, -22, 122, 102]                       # 100 x 40 grown to 125 x 50, squared up to 125 x 125
    kp = np.zeros((1, 4, 17, 2), np.float32)
    kp[0, 1, 3] = [np.nan, 50.]
    r = pc(dict(keypoint=kp, img_shape=(240, 320)))
    assert r['img_shape'] == (240, 320) and r['keypoint'][0, 1, 3].tolist() == [0., 50.]           # NaN -> 0, too narrow


def _compact_box_scalar(ext, hw, padding, hw_ratio, allow_imgpad, legacy):
    """The contract of augmentations.py:72-99 evaluated value by value with EXPLICIT precisions: ``legacy`` = NumPy 1.x
    scalar promotion (fp32 scalar op Python float -> float64), otherwise NEP 50 (the Python float adopts fp32)."""
    f32 = np.float32
    wide = np.float64 if legacy else np.float32
    lo_x, lo_y, hi_x, hi_y = (f32(v) for v in ext)
    cx, cy = f32(f32(hi_x + lo_x) / f32(2)), f32(f32(hi_y + lo_y) / f32(2))
    hw_, hh_ = wide(f32(f32(hi_x - lo_x) / f32(2))) * wide(1 + padding), wide(f32(f32(hi_y - lo_y) / f32(2))) * wide(1 + padding)
    if hw_ratio is not None:
        hh_ = max(wide(hw_ratio[0]) * hw_, hh_)
        hw_ = max(wide(1 / hw_ratio[1]) * hh_, hw_)
    x0, x1, y0, y1 = wide(cx) - hw_, wide(cx) + hw_, wide(cy) - hh_, wide(cy) + hh_
    if not allow_imgpad:
        x0, y0, x1, y1 = max(0, x0), max(0, y0), min(hw[1], x1), min(hw[0], y1)
    return [int(x0), int(y0), int(x1), int(y1)]


def test_compact_boxes_promotion_rules_on_boundary_values():
    """PoseCompact's scalar arithmetic promotes differently under NumPy 1.x (the reference's line: it uses
    np.Inf) and NumPy >= 2.  ``compact_boxes(promotion=)`` states which rule it follows; both are checked value by value
    against the explicit-precision evaluation above, on extents whose padded edges sit at / next to integers (where the
    int() truncation can land a pixel apart) and on random ones; the default is the reference's own line ('legacy')."""
    rng = np.random.default_rng(7)
    ext = []
    for _ in range(4000):
        lo = rng.uniform(0, 200, 2).astype(np.float32)
        span = rng.uniform(10, 300, 2).astype(np.float32)
        ext.append([lo[0], lo[1], lo[0] + span[0], lo[1] + span[1]])
    # edges constructed to hit integers after padding 0.25: span = 8k/5 * ... ; and fp32 neighbours of such values
    for k in range(16, 400, 7):
        base = np.float32(k)
        for d in (0, 1, -1):
            hi = np.nextafter(np.float32(base + 0.8 * k), np.float32(np.inf if d > 0 else -np.inf)) if d else np.float32(base + 0.8 * k)
            ext.append([base, base, hi, hi])
            ext.append([np.float32(base + 0.1), np.float32(base + 0.3), hi, np.float32(hi + 12.7)])
    ext = np.asarray(ext, np.float32)
    hw = [(240, 320)] * len(ext)
    differ = 0
    for padding, ratio, pad in ((0.25, None, True), (0.25, (1., 1.), True), (0.1, (0.75, 1.25), False), (1 / 3, None, False)):
        got = {}
        for rule in ('legacy', 'nep50'):
            apply, box = P.compact_boxes(ext, hw, padding, 10, ratio, pad, promotion=rule)
            assert apply.all()
            want = np.array([_compact_box_scalar(e, (240, 320), padding, ratio, pad, rule == 'legacy') for e in ext])
            assert np.array_equal(box, want), (rule, padding, ratio, pad, np.flatnonzero((box != want).any(1))[:5])
            got[rule] = box
        differ += int((got['legacy'] != got['nep50']).any(1).sum())
    assert differ > 0                                  # the boundary cases really separate the two rules
    assert np.array_equal(P.compact_boxes(ext, hw)[1], P.compact_boxes(ext, hw, promotion='legacy')[1])
    with pytest.raises(ValueError):
        P.compact_boxes(ext, hw, promotion='numpy')


def test_k400_dataset_builds_from_the_config_section(tmp_path):
    """config 5's `data.train` dict (PoseDataset with box_thr / valid_ratio over the compressed pickle + the pipeline above)
    builds through the registry and yields network inputs."""
    path = tmp_path / 'k400.pkl'
    with open(path, 'wb') as f:
        pickle.dump(annotations_k400(), f)
    ds = D.build_dataset(dict(type='PoseDataset', ann_file=str(path), pipeline=pipelines_k400()['k400_train'], box_thr=0.5,
                              valid_ratio=0.0))
    assert len(ds) == 4
    np.random.seed(0)
    item = ds[2]
    assert tuple(item['keypoint'].shape) == (1, 2, CLIP_LEN, 17, 3) and item['label'] == annotations_k400()[2]['label']
