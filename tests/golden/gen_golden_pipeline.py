"""Input-pipeline golden fixture from the IMPORTED reference transforms (build container only).

    python tests/golden/gen_golden_pipeline.py

For every pipeline config of pipeline_cases.pipelines() the reference's Compose (pyskl/datasets/pipelines) is run over the
seven synthetic clips with numpy's global RNG seeded once per config; the resulting network inputs and labels are
stored in pipeline.npz.  Data only."""
import copy
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim_data  # noqa: E402
from pipeline_cases import (annotations, annotations_2d, annotations_k400, pipelines, pipelines_2d,  # noqa: E402
                            pipelines_k400)

R = ref_shim_data.load()
out = {}
for pi, (name, cfg) in enumerate(pipelines().items()):
    pipe = R.Compose(copy.deepcopy(cfg))
    np.random.seed(1000 + pi)
    for si, ann in enumerate(annotations()):
        sample = copy.deepcopy(ann)
        sample.update(start_index=0, modality='Pose')
        res = pipe(sample)
        out[f'{name}_{si}'] = res['keypoint'].numpy()
        assert res['label'] == ann['label']
for pi, (name, cfg) in enumerate(pipelines_2d().items()):          # 2-D pose pickles: score channel, per-clip img_shape
    pipe = R.Compose(copy.deepcopy(cfg))
    np.random.seed(2000 + pi)
    for si, ann in enumerate(annotations_2d()):
        sample = copy.deepcopy(ann)
        sample.update(start_index=0, modality='Pose')
        res = pipe(sample)
        out[f'{name}_{si}'] = res['keypoint'].numpy()
# compressed Kinetics pose annotations: DecompressPose -> ... -> PoseCompact (box_thr 0.5: the dataset's anno_inds).  This
# container's scipy returns scalars from stats.mode; the reference indexes the old array form (`get_mode(...)[-1][0]`):
# give it that form (same count) instead of editing it
import scipy.stats  # noqa: E402
R.pose_related.get_mode = lambda a: scipy.stats.mode(a, keepdims=True)
if not hasattr(np, 'Inf'):
    np.Inf = np.inf          # (numpy 2 dropped the alias the reference's PoseCompact spells, augmentations.py:74-77)
for pi, (name, cfg) in enumerate(pipelines_k400().items()):
    pipe = R.Compose(copy.deepcopy(cfg))
    np.random.seed(3000 + pi)
    for si, ann in enumerate(annotations_k400()):
        sample = copy.deepcopy(ann)
        sample['anno_inds'] = sample.pop('box_score') >= 0.5
        sample.pop('valid')
        sample.update(start_index=0, modality='Pose')
        res = pipe(sample)
        out[f'{name}_{si}'] = res['keypoint'].numpy()
np.savez_compressed(os.path.join(HERE, 'pipeline.npz'), **out)
print('pipeline.npz', os.path.getsize(os.path.join(HERE, 'pipeline.npz')), len(out), 'arrays;',
      {k: v.shape for k, v in list(out.items())[:3]})
