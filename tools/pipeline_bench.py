"""Throughput of the batched input pipeline ("at >= 5 k clips/s/GPU the 8-worker numpy loader starves 8 GPUs"):
clips/s of SkeletonBatcher on the shipped NTU-60 training pipeline (configs/dsstgcn/ntu60_xsub_3dkp/j.py:11-20, clip_len 64
as in BASELINE), synthetic NTU-shaped clips resident in HBM.  Reports the host half (per-clip decisions, one Python thread),
the device half (dsgcn_skeleton_prep, HIP events) and the end-to-end rate, next to the per-sample host Compose (what ONE
loader worker of the reference does).      python tools/pipeline_bench.py [clips] [batch] [ntu|k400]
k400: BASELINE config 5's data section (configs/dsstgcn/kinetics400_hrnet/j.py:25-39) on synthetic compressed HRNet
detections (one row per detection, 1-4 persons per frame, T 150..300, clip_len 100): the store unpacks them once."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dsgcn_amd as D  # noqa: E402
from dsgcn_amd import pipeline as P  # noqa: E402

PIPE = [dict(type='PreNormalize3D', align_spine=False), dict(type='RandomRot', theta=0.2),
        dict(type='GenSkeFeat', feats=['j']), dict(type='UniformSample', clip_len=64), dict(type='PoseDecode'),
        dict(type='FormatGCNInput'), dict(type='Collect', keys=['keypoint', 'label'], meta_keys=[]),
        dict(type='ToTensor', keys=['keypoint'])]


def clips(n, seed=0):
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        T = int(rng.randint(50, 160))
        kp = (rng.randn(2, T, 25, 3) * 0.3 + np.array([0.1, 0.2, 3.0])).astype(np.float32)
        if i % 3:                       # most NTU clips hold one person
            kp[1] = 0
        out.append(dict(frame_dir=f'c{i}', label=i % 60, keypoint=kp, total_frames=T))
    return out


PIPE_K400 = [dict(type='DecompressPose', squeeze=True), dict(type='UniformSampleFrames', clip_len=100),
             dict(type='PoseDecode'), dict(type='PoseCompact', hw_ratio=1., allow_imgpad=True),
             dict(type='GenSkeFeat', dataset='coco', feats=['j']), dict(type='FormatGCNInput', num_person=2),
             dict(type='Collect', keys=['keypoint', 'label'], meta_keys=[]), dict(type='ToTensor', keys=['keypoint'])]


def clips_k400(n, seed=0):
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        T = int(rng.randint(150, 301))
        per = rng.randint(1, 5, size=T)
        per[rng.rand(T) < 0.05] = 0                     # frames without a detection
        fi = np.repeat(np.arange(T), per).astype(np.int16)
        D = len(fi)
        k = np.concatenate([rng.rand(D, 17, 2) * np.array([180, 120]) + np.array([60, 40]), rng.rand(D, 17, 1) * 0.6 + 0.3], 2)
        out.append(dict(frame_dir=f'k{i}', label=i % 400, img_shape=(240, 320), original_shape=(240, 320), total_frames=T,
                        frame_inds=fi, keypoint=k.astype(np.float16), anno_inds=rng.rand(D) < 0.9))
    return out


def main():
    global PIPE
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    kind = sys.argv[3] if len(sys.argv) > 3 else 'ntu'
    if kind == 'k400':
        PIPE = PIPE_K400
        anns = clips_k400(n)
        t0 = time.perf_counter()
        store = P.SkeletonStore(anns)
        print(f'store built (detections unpacked once, {store.data.numel() * 4 / 2**20:.0f} MiB resident): '
              f'{n / (time.perf_counter() - t0):.0f} clips/s')
    else:
        anns = clips(n)
        store = P.SkeletonStore(anns)
    batcher = P.SkeletonBatcher(PIPE)
    np.random.seed(0)
    order = np.random.permutation(n)
    batches = [order[i:i + B].tolist() for i in range(0, n - B + 1, B)]
    batcher(store, batches[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plans = [batcher.plan(store, b) for b in batches]
    t_first = time.perf_counter() - t0               # first visit: the per-clip geometry decisions are made (and cached)
    t0 = time.perf_counter()
    plans = [batcher.plan(store, b) for b in batches]
    t_plan = time.perf_counter() - t0                # every later epoch: RNG draws + gathers only
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for p in plans:
        out = batcher.run(store, p)
    e1.record()
    torch.cuda.synchronize()
    t_run = time.perf_counter() - t0
    t0 = time.perf_counter()
    for b in batches:
        out = batcher(store, b)
    torch.cuda.synchronize()
    t_e2e = time.perf_counter() - t0
    done = len(batches) * B
    comp = P.Compose(PIPE)
    t0 = time.perf_counter()
    k = min(256, n)
    for a in anns[:k]:
        s = dict(a, keypoint=a['keypoint'].copy(), start_index=0, modality='Pose')
        if 'frame_inds' in s:
            s['frame_inds'] = s['frame_inds'].copy()
        comp(s)
    t_host = time.perf_counter() - t0
    shape = 'NTU-shaped, T 50..160, clip_len 64' if kind != 'k400' else 'K400 HRNet detections, T 150..300, clip_len 100'
    print(f'{done} clips in batches of {B} ({shape} -> {tuple(out[0].shape)})')
    print(f'  host plan, first visit of a clip (geometry decisions made once, cached on the store): {done / t_first:9.0f} clips/s')
    print(f'  host plan, every later epoch (RNG draws clip by clip + batched gathers, 1 Python thread): {done / t_plan:9.0f} clips/s')
    print(f'  device half (H2D of the decisions + dsgcn_skeleton_prep), wall: {done / t_run:9.0f} clips/s; '
          f'HIP events {e0.elapsed_time(e1) / len(batches) * 1e3:.1f} us per batch')
    print(f'  end to end (plan + run, serial): {done / t_e2e:9.0f} clips/s')
    print(f'  per-sample host Compose (one loader worker of the reference\'s model): {k / t_host:9.0f} clips/s')


if __name__ == '__main__':
    main()
