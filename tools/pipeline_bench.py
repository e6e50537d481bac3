"""Throughput of the batched input pipeline ("at >= 5 k clips/s/GPU the 8-worker numpy loader starves 8 GPUs"):
clips/s of SkeletonBatcher on the shipped NTU-60 training pipeline (configs/dsstgcn/ntu60_xsub_3dkp/j.py:11-20, clip_len 64
as in BASELINE), synthetic NTU-shaped clips resident in HBM.  Reports the host half (per-clip decisions, one Python thread),
the device half (dsgcn_skeleton_prep, HIP events) and the end-to-end rate, next to the per-sample host Compose (what ONE
loader worker of the reference does).      python tools/pipeline_bench.py [clips] [batch]"""
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


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
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
        comp(s)
    t_host = time.perf_counter() - t0
    print(f'{done} clips in batches of {B} (NTU-shaped, T 50..160, clip_len 64 -> {tuple(out[0].shape)})')
    print(f'  host plan, first visit of a clip (geometry decisions made once, cached on the store): {done / t_first:9.0f} clips/s')
    print(f'  host plan, every later epoch (RNG draws clip by clip + batched gathers, 1 Python thread): {done / t_plan:9.0f} clips/s')
    print(f'  device half (H2D of the decisions + dsgcn_skeleton_prep), wall: {done / t_run:9.0f} clips/s; '
          f'HIP events {e0.elapsed_time(e1) / len(batches) * 1e3:.1f} us per batch')
    print(f'  end to end (plan + run, serial): {done / t_e2e:9.0f} clips/s')
    print(f'  per-sample host Compose (one loader worker of the reference\'s model): {k / t_host:9.0f} clips/s')


if __name__ == '__main__':
    main()
