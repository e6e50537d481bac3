"""Same-box A/B of the bench step under lab tuning keys: the lab library stands in for the product library, every variant
(comma-joined key=value pairs of dsgcn_pwconv_tuning; '' = defaults) gets a fresh model + TrainEngine (two hipGraphs) and is
timed over `steps` replays, the variants interleaved `rounds` times so that clock / box drift shows up as spread.
    python tools/step_ab.py '' 15=0 14=1 py:WSPLIT_BATCH=0 [--steps 20] [--rounds 2] [--kind ds|ctrgcn|stgcn|...]
(py:NAME=V sets the module-level switch NAME of ds-gcn_amd/kernels.py for that variant; tc:K=V a knob of dsgcn_tconv_tuning)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench, dsgcn_amd
from dsgcn_amd import native

args = sys.argv[1:]
steps, rounds, kind = 20, 2, 'ds'
variants = []
i = 0
while i < len(args):
    if args[i] == '--steps':
        steps = int(args[i + 1]); i += 2
    elif args[i] == '--rounds':
        rounds = int(args[i + 1]); i += 2
    elif args[i] == '--kind':
        kind = args[i + 1]; i += 2
    else:
        variants.append(args[i]); i += 1
variants = variants or ['']
lab = native.lab_lib()
native._lib = lab                                   # the package now launches through the lab build (same kernels + knobs)
DEFAULTS = {14: 2, 15: 3, 16: 256, 17: 128}
PY_DEFAULTS = {}
TC_DEFAULTS = {0: 1, 1: 1}
TC_SET = {}


def set_keys(variant):
    keys = dict(DEFAULTS)
    TC_SET.clear(); TC_SET.update(TC_DEFAULTS)
    from dsgcn_amd import kernels
    for name, val in PY_DEFAULTS.items():
        setattr(kernels, name, val)
    if variant:
        for kv in variant.split(','):
            k, v = kv.split('=')
            if k.startswith('tc:'):                     # a knob of the 9-tap family (dsgcn_tconv_tuning), e.g. tc:1=0
                TC_SET[int(k[3:])] = int(v)
                continue
            if k.startswith('py:'):                     # a module-level switch of ds-gcn_amd/kernels.py, e.g. py:WSPLIT_BATCH=0
                name = k[3:]
                PY_DEFAULTS.setdefault(name, getattr(kernels, name))
                old = PY_DEFAULTS[name]
                setattr(kernels, name, type(old)(int(v)) if isinstance(old, (bool, int)) else v)
                continue
            keys[int(k)] = int(v)
    for k, v in TC_SET.items():
        assert lab.dsgcn_tconv_tuning(k, v) == 0, (k, v)
    keys.setdefault(100, 1)
    for k, v in keys.items():
        if k == 100:                                    # 100 = fuse_out backward: 16-byte form on / off
            assert lab.dsgcn_fuse_out_tuning(0, v) == 0
        else:
            assert lab.dsgcn_pwconv_tuning(k, v) == 0, (k, v)


def run(variant):
    set_keys(variant)
    dev = torch.device('cuda')
    if kind == 'ds':
        model = bench.build_model().to(dev).train()
        T, V, classes, B = 64, 25, 60, 64
    else:
        np.random.seed(0); torch.manual_seed(0)
        model = dsgcn_amd.build_model(bench.other_cfg(kind)).to(dev).train()
        for mod in model.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        T, V, classes, B = 64, 25, 60, 64
    engine = dsgcn_amd.TrainEngine(model, lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True, use_graph=True, warmup_eager=2)
    g = torch.Generator().manual_seed(1234)
    kp = torch.randn(B, 1, 2, T, V, 3, generator=g).to(dev)
    lb = torch.randint(0, classes, (B, 1), generator=g).to(dev)
    for _ in range(5):
        engine.step(kp, lb)
    torch.cuda.synchronize()
    assert engine.graphed(kp, lb), engine.capture_error
    t0 = time.perf_counter()
    for _ in range(steps):
        engine.step(kp, lb)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    del engine, model
    torch.cuda.empty_cache()
    return ms


res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        res[v].append(run(v))
for v in variants:
    print(f'{v or "default":24s} ms/step: ' + ' '.join(f'{t:.3f}' for t in res[v]) + f'   min {min(res[v]):.3f}', flush=True)
