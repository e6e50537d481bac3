"""End-to-end throughput of ``train_model`` on the resident input pipeline: a synthetic NTU-shaped store
of `clips` clips in HBM, the shipped NTU-60 training pipeline (configs/dsstgcn/ntu60_xsub_3dkp/j.py:11-20, clip_len 64 as
in BASELINE), DS-STGCN, batch 64, hipGraph step — clips/s over whole epochs (pipeline + step + logging every 20
iterations), next to the bench replay of the same step on one fixed batch.
    python tools/train_bench.py [clips] [epochs]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import dsgcn_amd as D  # noqa: E402
from dsgcn_amd import pipeline as P  # noqa: E402
from dsgcn_amd.apis import train_model  # noqa: E402
from pipeline_bench import PIPE, clips  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    B = 64
    dev = torch.device('cuda')
    store = P.SkeletonStore(clips(n))
    res = {}
    for feeder in (True, False):
        batcher = P.SkeletonBatcher(PIPE)
        model = bench.build_model()
        cfg = dict(data=dict(videos_per_gpu=B, train_dataloader=dict(drop_last=True)), seed=0, total_epochs=epochs + 1,
                   optimizer=dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=5e-4, nesterov=True),
                   optimizer_config=dict(grad_clip=None), lr_config=dict(policy='CosineAnnealing', min_lr=0, by_epoch=False),
                   checkpoint_config=None, log_config=dict(interval=20), work_dir=None)
        # epoch 0 = warm-up (eager steps, graph capture, the per-clip decision cache); the timed part is the epochs after it
        import dsgcn_amd.apis as A
        marks = []
        orig = A.EpochRunner.train_epoch

        def timed(self):
            torch.cuda.synchronize()
            marks.append(time.perf_counter())
            orig(self)
            torch.cuda.synchronize()
            marks.append(time.perf_counter())
        A.EpochRunner.train_epoch = timed
        try:
            np.random.seed(0)
            runner = train_model(model, (store, batcher), cfg, device=dev, use_graph=True, prefetch=feeder)
        finally:
            A.EpochRunner.train_epoch = orig
        per_epoch = (n // B) * B
        dt = marks[-1] - marks[2]                        # epochs 1..: from the start of the second to the end of the last
        res[feeder] = per_epoch * epochs / dt
        assert runner.engine.graphed(torch.empty(B, 1, 2, 64, 25, 3), torch.empty(B, 1)), runner.engine.capture_error
        if feeder:
            kp, lb = batcher(store, list(range(B)))
            for _ in range(5):
                runner.engine.step(kp, lb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                runner.engine.step(kp, lb)
            torch.cuda.synchronize()
            replay = B * 30 / (time.perf_counter() - t0)
    print(f'train_model on a resident store of {n} clips, batch {B}, {epochs} timed epochs after a warm-up epoch')
    print(f'  end to end, feeder thread one batch ahead: {res[True]:8.0f} clips/s')
    print(f'  end to end, plan inline (no feeder thread): {res[False]:8.0f} clips/s')
    print(f'  bench replay of the same step on one fixed batch: {replay:8.0f} clips/s  -> end to end = {res[True] / replay:.3f} of it')


if __name__ == '__main__':
    main()
