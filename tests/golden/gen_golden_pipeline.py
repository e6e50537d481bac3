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
from pipeline_cases import annotations, annotations_2d, pipelines, pipelines_2d  # noqa: E402

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
np.savez_compressed(os.path.join(HERE, 'pipeline.npz'), **out)
print('pipeline.npz', os.path.getsize(os.path.join(HERE, 'pipeline.npz')), len(out), 'arrays;',
      {k: v.shape for k, v in list(out.items())[:3]})
