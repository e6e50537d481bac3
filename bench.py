#!/usr/bin/env python3
"""bench.py — clips/sec forward+backward, DS-STGCN NTU-60 (3x64x25x2 clips), on N MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N>1 it is launched through
``torch.distributed.run`` (one rank per GPU, RCCL) — by the driver, or, when called plainly with N>1, by itself
(``self_launch``: a child launcher started before anything touches the GPU).  Rank 0 prints ONE JSON line.

A "step" = one pass of the hot path over one batch of synthetic clips resident in HBM: forward (train-mode
BatchNorm, CE loss) + backward + gradient all-reduce (N>1) + the SGD-nesterov update.  Weak scaling: 64
clips per GPU.  The line also carries
  roofline     — the gather-aggregate kernel (K-A) timed live with HIP events on the launch stream over the
                 model's own 10-layer shape mix, against the 8 TB/s HBM peak (algorithmic bytes only);
  cpu_baseline — the CPU oracle (op-for-op PyTorch restatement of the reference path) timed on this box's
                 host cores on a bounded sample of the same workload (rank 0, N=1 only);
  roofline_step — the whole step against the HBM peak: algorithmic bytes from the layer table below, the committed PMC
                 total of a step, and this run's ms_per_step;
  other_configs — BASELINE configs 1 / 3 / 4 / 5 (per GPU) through the same engine, 10 hipGraph steps each (N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured copy)
CLIPS_PER_GPU = 64
T, V, M, C, CLASSES = 64, 25, 2, 3, 60


def ds_cfg(num_classes=CLASSES, layout='nturgb+d'):
    return dict(
        type='RecognizerGCN',
        backbone=dict(
            type='DGSTGCN', gcn_type='dgphgcn1', gcn_ratio=0.125, gcn_node_attention=True, gcn_edge_attention=True,
            gcn_decompose=True, gcn_subset_wise=True, gcn_ctr='T', gcn_ada='T', tcn_type='dgmstcn',
            graph_cfg=dict(layout=layout, mode='random', num_filter=3, init_off=.04, init_std=.02),
            tcn_ms_cfg=[(3, 1), (3, 2), (3, 3), (3, 4), ('max', 3), '1x1']),
        cls_head=dict(type='GCNHead', num_classes=num_classes, in_channels=256))


def other_cfg(kind, num_classes=60, **bk):
    """Model dicts of the other backbones that run on the same kernels: 'stgcn' (vanilla ST-GCN, BASELINE config 1),
    'stgcnpp' (configs/stgcn++), 'ctrgcn' (classic CTR-GCN, BASELINE config 4), 'aagcn', 'dggcn' (the f-4 backbones)."""
    if kind == 'ctrgcn':
        backbone = dict(type='CTRGCN', gcn_type='unit_ctrgcn', graph_cfg=dict(layout='nturgb+d', mode='spatial'))
    elif kind == 'ctrgcn_shipped':        # configs/ctrgcn/CTRGCN_model.py: unit_ctrhgcn + msmlp on the random graph
        backbone = dict(type='CTRGCN', gcn_type='unit_ctrhgcn', gcn_node_attention=True, gcn_edge_attention=True,
                        gcn_add_type=False, gcn_ada=True, gcn_num_types=5, gcn_rel_reduction=8, gcn_edge_num=15,
                        tcn_type='msmlp', tcn_add_tcn=True, tcn_merge_after=True,
                        graph_cfg=dict(layout='nturgb+d', mode='random', num_filter=3, init_off=.04, init_std=.02))
    elif kind == 'aagcn':                 # 2s-AGCN / AAGCN with its class defaults (unit_aagcn + unit_tcn k=9)
        backbone = dict(type='AAGCN', graph_cfg=dict(layout='nturgb+d', mode='spatial'))
    elif kind == 'dggcn':                 # the original DG-STGCN (PYSKL dgstgcn configs): dggcn + dgmstcn on the random graph
        backbone = dict(type='DGSTGCN', gcn_type='dggcn', gcn_ratio=0.125, gcn_ctr='T', gcn_ada='T', tcn_type='dgmstcn',
                        graph_cfg=dict(layout='nturgb+d', mode='random', num_filter=3, init_off=.04, init_std=.02),
                        tcn_ms_cfg=[(3, 1), (3, 2), (3, 3), (3, 4), ('max', 3), '1x1'])
    elif kind == 'stgcn_shipped':         # configs/stgcn/STGCN_model.py: unit_gcn + unitmlp (k=9) on the random graph
        backbone = dict(type='STGCN', gcn_adaptive='init', tcn_type='unitmlp', tcn_add_tcn=True, tcn_merge_after=True,
                        graph_cfg=dict(layout='nturgb+d', mode='random', num_filter=3, init_off=.04, init_std=.02))
    elif kind == 'stgcnpp':
        backbone = dict(type='STGCN', gcn_adaptive='init', gcn_with_res=True, tcn_type='mstcn',
                        graph_cfg=dict(layout='nturgb+d', mode='spatial'))
    else:
        backbone = dict(type='STGCN', graph_cfg=dict(layout='nturgb+d', mode='stgcn_spatial'))
    backbone.update(bk)
    return dict(type='RecognizerGCN', backbone=backbone,
                cls_head=dict(type='GCNHead', num_classes=num_classes, in_channels=256))


def build_model(seed=0):
    import dsgcn_amd
    np.random.seed(seed)
    torch.manual_seed(seed)
    model = dsgcn_amd.build_model(ds_cfg())
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():   # zero-init alpha/beta/add_coeff would switch the dynamic paths off
        for k, p in model.named_parameters():
            if k.endswith(('alpha', 'beta', 'add_coeff')):
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)
    return model


def ka_layer_shapes(n):
    """(n, K*mid, T_at_gcn) of the 10 DS-STGCN layers."""
    return [(n, 24, 64)] * 4 + [(n, 48, 64)] + [(n, 48, 32)] * 2 + [(n, 96, 32)] + [(n, 96, 16)] * 2


def ka_alg_bytes(n, KC, t, v, bwd):
    units = n * KC
    return 4 * units * ((3 * t * v + 2 * v * v) if bwd else (2 * t * v + v * v))


def _pmc_traffic(kernel):
    """(HBM bytes per launch, where the figure comes from): the PMC counters cannot be read from inside this process, so
    `traffic` is the latest COMMITTED rocprofv3 --pmc result (two passes, FETCH_SIZE x2 per the gfx950 correction +
    WRITE_SIZE, tools/ka_once.py as the profiled program, tools/gpu/ka_pmc.sh) — not a measurement of this run, and
    `traffic_source` says so; (None, None) when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', 'ka_traffic.json')))
    if not files:
        return None, None
    with open(files[-1]) as f:
        d = json.load(f)
    for k, v in d.items():
        if k.startswith(kernel):
            return v.get('hbm_bytes_per_launch'), ('committed PMC passes ' + os.path.relpath(files[-1], ROOT) +
                                                   ' (rocprofv3 --pmc, separate run of the same kernels; not measured by this process)')
    return None, None


def _in_step_frac(prefix, n, bwd):
    """The same kernels INSIDE the replayed step: (frac, avg us, source) from the latest committed step sequence
    (profiles/rNN/step_sequence.txt: the kernels of one graph replay in launch order, rocprofv3 --kernel-trace of
    `bench.py` — tools/step_sequence.py).  Like `traffic`: a committed profile of this code, not a measurement of this run.
    Behind the kernels that precede them in the step (dirty lines, cold operands) the launches run a few percent slower
    than in the isolated loop `frac` is measured on."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', 'step_sequence.txt')))
    if not files:
        return None, None, None
    us = []
    with open(files[-1]) as f:
        for line in f:
            m = re.match(r'\s*([\d.]+) us\s+(\S+)', line)
            if m and m.group(2).startswith(prefix):
                us.append(float(m.group(1)))
    shapes = ka_layer_shapes(n)
    if len(us) != len(shapes):
        return None, None, None
    nbytes = sum(ka_alg_bytes(*sh, V, bwd) for sh in shapes)
    total = sum(us)
    return (round(nbytes / (total * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), round(total / len(us), 2),
            'committed graph-replay kernel trace ' + os.path.relpath(files[-1], ROOT) + ' (not measured by this process)')


def measure_ka_roofline(device, n, reps=20, blocks=5):
    """HIP-event timing of K-A fwd and bwd over the model's layer mix (distinct buffers per layer so the
    working set, 0.64 GB fwd / 1.05 GB bwd, exceeds the 256 MiB Infinity Cache).  `blocks` timed blocks of `reps`
    repetitions each: `achieved` / `frac` / `avg_launch_us` are the MEDIAN block, `frac_min` / `frac_max` the spread (a
    26 us launch moves by a few percent between blocks and between boxes of the pool)."""
    from dsgcn_amd import native
    lib = native.lib()
    st = torch.cuda.current_stream().cuda_stream
    bufs = []
    for (nn, KC, t) in ka_layer_shapes(n):
        zp = torch.randn(nn, KC, t, V, device=device)
        ah = torch.randn(nn, KC, V, V, device=device) * 0.2
        sc = torch.rand(KC, device=device) + 0.5
        sh = torch.randn(KC, device=device) * 0.1
        bufs.append(dict(zp=zp, ah=ah, sc=sc, sh=sh, y=torch.empty_like(zp), dy=torch.randn_like(zp),
                         dzp=torch.empty_like(zp), dah=torch.empty_like(ah),
                         part=torch.empty(4 * nn * KC, 2, device=device), dims=(nn, KC, t)))

    def fwd(b):
        nn, KC, t = b['dims']
        rc = lib.dsgcn_aggregate_fwd(b['zp'].data_ptr(), b['sc'].data_ptr(), b['sh'].data_ptr(), 1, b['ah'].data_ptr(),
                                     b['y'].data_ptr(), nn, KC, t, V, st)
        assert rc == 0, rc

    def bwd(b):
        nn, KC, t = b['dims']
        rc = lib.dsgcn_aggregate_bwd(b['zp'].data_ptr(), b['sc'].data_ptr(), b['sh'].data_ptr(), 1, b['ah'].data_ptr(),
                                     b['dy'].data_ptr(), b['dzp'].data_ptr(), b['dah'].data_ptr(),
                                     b['part'].data_ptr(), nn, KC, t, V, st)
        assert rc == 0, rc

    out = {}
    for name, fn, is_bwd in (('k_aggregate_fwd', fwd, False), ('k_aggregate_bwd', bwd, True)):
        for b in bufs:
            fn(b)
        torch.cuda.synchronize()
        times = []
        for _ in range(blocks):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                for b in bufs:
                    fn(b)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / reps)
        times.sort()
        ms = times[len(times) // 2]
        nbytes = sum(ka_alg_bytes(*b['dims'], V, is_bwd) for b in bufs)
        launches = len(bufs)
        gbs = nbytes / (ms * 1e-3) / 1e9
        traffic, src = _pmc_traffic(name)
        isf, isus, issrc = _in_step_frac(name, n, is_bwd)
        out[name] = dict(bound='hbm', achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                         frac=round(gbs / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=src,
                         in_step_frac=isf, in_step_avg_launch_us=isus, in_step_source=issrc,
                         avg_launch_us=round(ms * 1e3 / launches, 2), alg_bytes_per_launch=nbytes // launches,
                         launches_per_step=launches, blocks=blocks, reps_per_block=reps,
                         frac_min=round(nbytes / (times[-1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         frac_max=round(nbytes / (times[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
    return out


def ctr_layer_shapes(n):
    """(n, Co, T at the spatial unit) of the 10 CTR-GCN blocks (BASELINE config 4: 64/64/64/64/128/128/128/256/256/256
    channels, temporal stride 2 in blocks 5 and 8)."""
    return [(n, 64, 64)] * 4 + [(n, 128, 64)] + [(n, 128, 32)] * 2 + [(n, 256, 32)] + [(n, 256, 16)] * 2


def measure_kap_roofline(device, n, reps=10, blocks=3, K=3):
    """K-A' (csrc/aggsum.hip: y[n,c] = sum_k P[n,k,c] . Ahat[n,k,c], the CTR-GCN gather-aggregate with its per-sample,
    per-channel adjacency) over CTR-GCN's own layer mix, HIP-event timed like K-A.  Algorithmic bytes per (n, c) plane
    (SURVEY §8 a10): forward 4*((K+1)*T*V + K*V*V), backward 4*((2K+1)*T*V + 2*K*V*V) (P_k, G in; dP_k, dAhat_k out)."""
    from dsgcn_amd import native
    lib = native.lib()
    st = torch.cuda.current_stream().cuda_stream
    bufs = []
    for (nn, Co, t) in ctr_layer_shapes(n):
        p = torch.randn(nn, K * Co, t, V, device=device)
        ah = torch.randn(nn, K * Co, V, V, device=device) * 0.2
        bufs.append(dict(p=p, ah=ah, y=torch.empty(nn, Co, t, V, device=device), gy=torch.randn(nn, Co, t, V, device=device),
                         dp=torch.empty_like(p), dah=torch.empty_like(ah),
                         part=torch.empty(lib.dsgcn_aggsum_partial_rows(nn, t, V), Co, 2, device=device), dims=(nn, Co, t)))

    def fwd(b):
        nn, Co, t = b['dims']
        rc = lib.dsgcn_aggsum_fwd(b['p'].data_ptr(), b['ah'].data_ptr(), K * Co * V * V, Co * V * V, V * V, b['y'].data_ptr(),
                                  b['part'].data_ptr(), nn, K, Co, t, V, st)
        assert rc == 0, rc

    def bwd(b):
        nn, Co, t = b['dims']
        rc = lib.dsgcn_aggsum_bwd(b['p'].data_ptr(), b['ah'].data_ptr(), K * Co * V * V, Co * V * V, V * V, b['gy'].data_ptr(),
                                  None, None, None, b['dp'].data_ptr(), b['dah'].data_ptr(), K * Co * V * V, Co * V * V, V * V,
                                  nn, K, Co, t, V, st)
        assert rc == 0, rc

    out = {}
    for name, fn, is_bwd in (('k_aggsum_fwd_ctr', fwd, False), ('k_aggsum_bwd_ctr', bwd, True)):
        for b in bufs:
            fn(b)
        torch.cuda.synchronize()
        times = []
        for _ in range(blocks):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                for b in bufs:
                    fn(b)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / reps)
        times.sort()
        ms = times[len(times) // 2]
        nbytes = sum(4 * nn * Co * (((2 * K + 1) if is_bwd else (K + 1)) * t * V + (2 if is_bwd else 1) * K * V * V)
                     for nn, Co, t in (b['dims'] for b in bufs))
        gbs = nbytes / (ms * 1e-3) / 1e9
        out[name] = dict(bound='hbm', achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(gbs / HBM_PEAK_GBS, 4),
                         traffic=None, avg_launch_us=round(ms * 1e3 / len(bufs), 2), alg_bytes_per_launch=nbytes // len(bufs),
                         launches_per_step=len(bufs), workload='CTR-GCN (BASELINE config 4) layer mix, K = 3, per-sample adjacency')
    return out


MFMA_F32_PEAK_TF = 157.3        # MI355X_MICROARCH.md: f32-input MFMA = vector f32 rate
MFMA_BF16_PEAK_TF = 2500.0      # MI355X_MICROARCH.md: dense bf16 MFMA


def measure_kc_roofline(device, n, reps=20, nsets=4):
    """K-C (1x1 channel mix) forward / data gradient / weight gradient through the C ABI, HIP-event timed, on the two
    shapes that bracket the model: 64->64 at T=64 (fp32 MFMA, 16 FLOP/B) and 256->256 at T=16 (six bf16 MFMA terms per
    fp32 product, csrc/common.h b3_split).  Every launch works on one of `nsets` disjoint operand sets (0.6-0.8 GB in
    rotation, beyond the 256 MiB Infinity Cache): HBM-cold operands, as in the step.
    The roof of a launch is max(algorithmic bytes / 8 TB/s, ISSUED matrix flops / the peak of the instruction that
    carries them): `frac` = roof time / measured time, `bound` names the larger term.  Algorithmic bytes: inputs read once
    + outputs written once.  `fp32_equiv_tflops` (2*Ci*Co per position, whatever carries it) is context only."""
    from dsgcn_amd import native
    lib = native.lib()
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    P = lambda tt: tt.data_ptr()      # noqa: E731
    for Ci, Co, t in ((64, 64, 64), (256, 256, 16)):
        s1 = torch.rand(Ci, device=device) + .5
        h1 = torch.randn(Ci, device=device) * .1
        w = torch.randn(Co, Ci, device=device) * Ci ** -.5
        b = torch.zeros(Co, device=device)
        A0 = torch.randn(Co, device=device) * 1e-3
        B0 = torch.randn(Co, device=device) * 1e-3
        part = torch.empty(lib.dsgcn_pwconv_partial_rows(n, Ci, Co, t, V, 1, 0), Co, 2, device=device)
        ipart = torch.empty(lib.dsgcn_pwconv_ipart_rows(n, Ci, Co, t, V, 1), Ci, 3, device=device)
        splits = lib.dsgcn_pwconv_wgrad_splits(n, Ci, Co, t, V, 1)
        pstride = Co * Ci + Co
        wpart = torch.empty(splits, pstride, device=device)
        sets = [dict(x=torch.randn(n, Ci, t, V, device=device), z=torch.randn(n, Co, t, V, device=device),
                     gz=torch.randn(n, Co, t, V, device=device), dx=torch.empty(n, Ci, t, V, device=device))
                for _ in range(nsets)]
        rows_f = lib.dsgcn_pwconv_bwd_rows(n, Ci, Co, t, V, 1)
        wpf = torch.empty(max(rows_f, 1), pstride, device=device)
        ipf = torch.empty(max(rows_f, 1), Ci, 3, device=device)

        # as the product calls them: the wide convs with their pre-split weight image (prepared once, outside the timing)
        wsb = lib.dsgcn_pwconv_wsplit_bytes(n, Ci, Co, t, V, 1)
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=device)
        if wsb:
            assert lib.dsgcn_pwconv_wsplit(P(w), Ci, Co, P(ws), st) == 0
        wsp = P(ws) if wsb else None

        def fwd(q):
            assert lib.dsgcn_pwconv_fwd_ws(P(q['x']), P(s1), P(h1), None, None, None, 1, P(w), P(b), P(q['z']), None,
                                           P(part), n, Ci, Co, t, V, 1, 0, 1, wsp, st) == 0

        def dgrad(q):
            assert lib.dsgcn_pwconv_dgrad_ws(P(q['x']), P(s1), P(h1), None, None, None, 1, P(w), P(q['z']), None,
                                             P(q['gz']), None, P(A0), P(B0), P(q['dx']), None, P(ipart), n, Ci, Co, t, V,
                                             1, 0, wsp, st) == 0

        def wgrad(q):
            assert lib.dsgcn_pwconv_wgrad(P(q['x']), P(s1), P(h1), None, None, None, 1, P(q['z']), None, P(q['gz']), None,
                                          P(A0), P(B0), wpart.data_ptr(), wpart.data_ptr() + 4 * Co * Ci, pstride, n, Ci,
                                          Co, t, V, 1, 0, st) == 0

        def bwd(q):
            assert lib.dsgcn_pwconv_bwd(P(q['x']), P(s1), P(h1), None, None, None, 1, P(w), P(q['z']), P(q['gz']), P(A0),
                                        P(B0), P(q['dx']), None, P(ipf), wpf.data_ptr(), wpf.data_ptr() + 4 * Co * Ci,
                                        pstride, n, Ci, Co, t, V, st) == 0
        L = n * t * V
        flops = 2.0 * Ci * Co * L
        jobs = [('fwd', fwd, 4 * L * (Ci + Co), 1), ('dgrad', dgrad, 4 * L * (2 * Co + 2 * Ci), 1),
                ('wgrad', wgrad, 4 * L * (2 * Co + Ci), 1)]
        if rows_f:        # narrow convs: the step runs both gradients as one pass (csrc/bwd64.hip)
            jobs.append(('bwd', bwd, 4 * L * (2 * Co + 2 * Ci), 2))
        wide = Ci > 64          # >= 128 channels: the bf16-term kernels (k_pwg, k_wg2<..., B3>)
        for name, fn, nbytes, nprod in jobs:
            for q in sets:
                fn(q)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(reps):
                fn(sets[r % nsets])
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            gbs, tf = nbytes / us / 1e3, nprod * flops / us / 1e6
            issued_tf = (6 if wide else 1) * tf
            t_hbm = nbytes / (HBM_PEAK_GBS * 1e3)                                        # us
            t_mfma = nprod * flops * (6 if wide else 1) / ((MFMA_BF16_PEAK_TF if wide else MFMA_F32_PEAK_TF) * 1e6)
            hbm_bound = t_hbm >= t_mfma
            row = dict(
                bound='hbm' if hbm_bound else 'mfma', achieved=round(gbs if hbm_bound else issued_tf, 1),
                peak=HBM_PEAK_GBS if hbm_bound else (MFMA_BF16_PEAK_TF if wide else MFMA_F32_PEAK_TF),
                unit='GB/s' if hbm_bound else 'TFLOP/s', frac=round(max(t_hbm, t_mfma) / us, 4), traffic=None,
                avg_launch_us=round(us, 2), roof_us=round(max(t_hbm, t_mfma), 2), hbm_gbs=round(gbs, 1),
                hbm_frac=round(gbs / HBM_PEAK_GBS, 4), fp32_equiv_tflops=round(tf, 1),
                mfma_instr='v_mfma_f32_32x32x16_bf16 x6 terms' if wide else 'v_mfma_f32_32x32x2_f32',
                issued_tflops=round(issued_tf, 1),
                mfma_util=round(issued_tf / (MFMA_BF16_PEAK_TF if wide else MFMA_F32_PEAK_TF), 4),
                operands=f'{nsets} rotating sets, HBM-cold')
            out[f'k_pwconv_{name}_{Ci}x{Co}'] = row
    return out


DS_PLAN = [(3, 64, 64, 1), (64, 64, 64, 1), (64, 64, 64, 1), (64, 64, 64, 1), (64, 128, 64, 2), (128, 128, 32, 1),
           (128, 128, 32, 1), (128, 256, 32, 2), (256, 256, 16, 1), (256, 256, 16, 1)]     # (Ci, Co, T at the block, stride)


def step_algorithmic_elements(plan=DS_PLAN, v=V):
    """The tensors of one DS-STGCN training step that HAVE to exist in HBM, per person-sample and per family:
    every tensor that sits behind a train-mode BatchNorm's global reduction (the raw conv outputs z), the dynamic adjacency,
    the gather-aggregate's output (the north_star's kernel boundary) and the block outputs.  Elements, not bytes."""
    rows = {'input': 3 * plan[0][2] * v}

    def add(k, e):
        rows[k] = rows.get(k, 0) + e
    for i, (ci, co, t, s) in enumerate(plan):
        mid, L = co // 8, t * v
        Lo = L // s
        add('z_pre', 3 * mid * L)              # pre conv out, BN + ReLU applied by K-A while loading
        add('Ahat', 3 * mid * v * v)           # K-B out
        add('Y', 3 * mid * L)                  # K-A out
        add('z_post', co * L)                  # post conv out (BN + residual + ReLU applied by the branch convs' load)
        if ci != co:
            add('z_down', co * L)              # the spatial unit's residual conv + BN
        add('z_branch', co * L)                # the six branch 1x1 convs (BN + ReLU applied by the window kernels' load)
        add('f', co * Lo)                      # windows + global-joint combine (transform.0 BN + ReLU applied by the next load)
        add('z_transform', co * Lo)            # transform conv out (BN applied by fuse_out)
        if s != 1:
            add('z_res', co * Lo)              # the block residual's strided 1x1 conv + BN
        if i + 1 < len(plan):
            add('out', co * Lo)                # block output (the last block's is only ever pooled: its plane means suffice)
    return rows


def step_roofline(n, ms_per_step):
    """Whole-step roofline.  Algorithmic bytes = 4 B x 5 touches x the tensors of step_algorithmic_elements:
    written once and read once forward, read once more in the backward, its gradient written once and read once.  `traffic` =
    the HBM bytes of one step from the latest COMMITTED PMC passes (profiles/rNN/step_hbm_traffic.csv, FETCH_SIZE x2 +
    WRITE_SIZE over an eager step of this same workload)."""
    import glob
    rows = step_algorithmic_elements()
    per_family = {k: 20 * n * e for k, e in rows.items()}
    alg = sum(per_family.values())
    out = dict(bound='hbm', alg_bytes=alg, alg_bytes_by_family={k: round(b / 1e9, 3) for k, b in per_family.items()},
               alg_unit='GB per step per family; 4 B x 5 touches (fwd write + read, bwd read, gradient write + read)',
               achieved=round(alg / (ms_per_step * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS, unit='GB/s',
               alg_frac=round(alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), traffic=None, traffic_ratio=None)
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', 'step_hbm_traffic.csv')))
    if files:
        import csv
        with open(files[-1]) as f:
            for row in csv.reader(f):
                if row and row[0] == 'TOTAL':
                    traffic = (float(row[2]) + float(row[3])) * 2 ** 20       # the csv's 'MB' are MiB (counter KB / 1024)
                    out.update(traffic=int(traffic), traffic_ratio=round(traffic / alg, 3),
                               traffic_frac=round(traffic / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               traffic_source='committed PMC passes ' + os.path.relpath(files[-1], ROOT) +
                                              ' (rocprofv3 --pmc over an eager step; not measured by this process)')
    return out


OTHER_CONFIGS = (('stgcn', 'BASELINE config 1: ST-GCN NTU-60 (vanilla, stgcn_spatial graph)', 64),
                 ('ds120', 'BASELINE config 3 per GPU: DS-STGCN NTU-120', 64),
                 ('ctrgcn', 'BASELINE config 4: CTR-GCN NTU-60 (unit_ctrgcn + MSTCN)', 64),
                 ('ds_k400', 'BASELINE config 5 per GPU: DS-STGCN Kinetics-400 2D keypoints (V=17, T=100)', 32))


def measure_other_configs(device, steps=10, warmup=3):
    """The other BASELINE configurations through the same TrainEngine step (fwd + bwd + SGD-nesterov, two hipGraphs), `steps`
    timed steps each on synthetic clips of their own shape — so that the driver's clock sees them too."""
    import gc
    import dsgcn_amd
    out = {}
    # (kind, workload, clips, dropout kept?): ST-GCN a second time with its shipped tcn_dropout = 0.5
    # (configs/stgcn/stgcn_vanilla_ntu60_xsub_3dkp/j.py:5) — fused into fuse_out since round 6, graph-capturable
    runs = [(k, w, c, False) for k, w, c in OTHER_CONFIGS]
    runs.insert(1, ('stgcn', 'BASELINE config 1 with its shipped training dropout (tcn_dropout=0.5, fused into fuse_out)',
                    dict((k, c) for k, _, c in OTHER_CONFIGS)['stgcn'], True))
    for kind, what, clips, keep_drop in runs:
        t_, v_, classes = 64, 25, 60
        if kind == 'ds_k400':
            cfg, t_, v_, classes = ds_cfg(400, 'coco'), 100, 17, 400
        elif kind == 'ds120':
            cfg, classes = ds_cfg(120), 120
        elif keep_drop:
            cfg = other_cfg(kind, tcn_dropout=0.5)
        else:
            cfg = other_cfg(kind)
        key = kind + '_dropout' if keep_drop else kind
        try:
            np.random.seed(0)
            torch.manual_seed(0)
            m = dsgcn_amd.build_model(cfg)
            gen = torch.Generator().manual_seed(1)
            with torch.no_grad():
                for k, p in m.named_parameters():
                    if k.endswith(('alpha', 'beta', 'add_coeff')):
                        p.copy_(torch.randn(p.shape, generator=gen) * 0.5)
            for mod in m.modules():
                if isinstance(mod, torch.nn.Dropout) and not keep_drop:
                    mod.p = 0.0
            m = m.to(device).train()
            eng = dsgcn_amd.TrainEngine(m, lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True, use_graph=True,
                                        warmup_eager=2, strict_graph=False)
            x = torch.randn(clips, 1, M, t_, v_, C, generator=gen).to(device)
            y = torch.randint(0, classes, (clips, 1), generator=gen).to(device)
            for _ in range(max(warmup, 3)):
                eng.step(x, y)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = eng.step(x, y)['loss']
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            out[key] = dict(workload=what, clips_per_gpu=clips, ms_per_step=round(dt * 1e3, 3), clips_per_s=round(clips / dt, 1),
                            steps=steps, hip_graph=bool(eng.graphed(x, y)), final_loss=round(float(loss.item()), 5),
                            step='fwd+bwd+SGD-nesterov (TrainEngine), dropout p=' + ('0.5' if keep_drop else '0'))
            del eng, m, x, y
        except Exception as exc:        # a secondary figure must never take the bench line down
            out[key] = dict(workload=what, error=f'{type(exc).__name__}: {exc}')
        gc.collect()
        torch.cuda.empty_cache()
    return out


def _cpu_topology():
    """(physical cores, hw threads, model name) from /proc/cpuinfo."""
    cores, threads, model, phys, core = set(), 0, 'unknown CPU', None, None
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('processor'):
                    threads += 1
                elif line.startswith('model name') and model == 'unknown CPU':
                    model = line.split(':', 1)[1].strip()
                elif line.startswith('physical id'):
                    phys = line.split(':', 1)[1].strip()
                elif line.startswith('core id'):
                    core = line.split(':', 1)[1].strip()
                    cores.add((phys, core))
    except OSError:
        pass
    threads = threads or (os.cpu_count() or 1)
    return (len(cores) or threads), threads, model


def cpu_baseline(budget_s=12.0, batch=16):
    """Oracle (CPU PyTorch restatement) fwd+bwd on this box's host cores: bounded sample.
    Intra-op threads are chosen by a short probe (8/16/32): with hundreds of tiny ATen ops per step more
    threads are SLOWER (measured on the 2x64-core EPYC host: 16 threads 21 clips/s, 128 threads 0.9)."""
    from oracle import dsgcn_oracle as O
    model = build_model()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    leaves = {k: v.requires_grad_() for k, v in sd.items() if v.dtype.is_floating_point and 'running' not in k}
    sd.update(leaves)
    gc = O.graph_constants('nturgb+d')
    plan = O.dgstgcn_plan()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(batch, 1, M, T, V, C, generator=g)
    y = torch.randint(0, CLASSES, (batch, 1), generator=g)

    def step():
        for v in leaves.values():
            v.grad = None
        _, loss = O.recognizer_forward_train(x, y, sd, gc['node_type'], gc['edge_type'], plan)
        loss.backward()

    avail = os.cpu_count() or 1
    best_thr, best_t = 1, float('inf')
    for thr in [c for c in (8, 16, 32) if c <= avail] or [avail]:
        torch.set_num_threads(thr)
        step()   # warm-up (allocator, mkldnn primitive cache)
        t0 = time.perf_counter()
        step()
        dt = time.perf_counter() - t0
        if dt < best_t:
            best_thr, best_t = thr, dt
    torch.set_num_threads(best_thr)
    t0 = time.perf_counter()
    iters = 0
    while True:
        step()
        iters += 1
        el = time.perf_counter() - t0
        if (el >= budget_s and iters >= 2) or iters >= 50 or el > 4 * budget_s:
            break
    el = time.perf_counter() - t0
    phys, hw, cpu_model = _cpu_topology()
    res = dict(value=round(batch * iters / el, 2), unit='clips/s', cores=best_thr, kind='port',
               sample=f'{iters} fwd+bwd iterations of a {batch}-clip batch (3x{T}x{V}x{M}), oracle/dsgcn_oracle.py, '
                      f'torch {torch.__version__} CPU, {best_thr} intra-op threads (best of 8/16/32); host: {cpu_model}, '
                      f'{phys} physical cores, {hw} hw threads (SMT {"on" if hw > phys else "off"}), {el:.1f} s')
    # The BASELINE batch itself (N = 64) at the best thread count: the 16-clip figure above keeps the sample
    # bounded; this one says what the CPU path does on the very batch the GPU step is timed on (one timed iteration when the
    # 16-clip rate says it fits 0.6 of the budget, else skipped and said so; measured 3x slower per clip than 16-clip batches).  
    try:
        est = 64.0 / max(res['value'], 1e-9)
        if est <= 0.6 * budget_s:
            g64 = torch.Generator().manual_seed(1234)
            xb = torch.randn(64, 1, M, T, V, C, generator=g64)
            yb = torch.randint(0, CLASSES, (64, 1), generator=g64)

            def step64():
                for v in leaves.values():
                    v.grad = None
                _, loss = O.recognizer_forward_train(xb, yb, sd, gc['node_type'], gc['edge_type'], plan)
                loss.backward()
            step64()
            t0 = time.perf_counter()
            step64()
            t64 = time.perf_counter() - t0
            res['batch64'] = dict(value=round(64 / t64, 2), unit='clips/s', cores=best_thr,
                                  sample=f'1 fwd+bwd iteration (after one warm-up) of the 64-clip BASELINE batch at {best_thr} '
                                         f'threads, {t64:.1f} s')
            del xb, yb
        else:
            res['batch64'] = dict(value=None, note=f'one 64-clip iteration would take ~{est:.0f} s at the 16-clip rate: not run')
    except Exception as exc:
        res['batch64'] = dict(value=None, error=f'{type(exc).__name__}: {exc}')
    # SURVEY §8(d) also asks for the BASELINE batch (N=64) on ALL physical cores.  With hundreds of small ATen ops per
    # step that configuration is far slower than the best thread count; it is timed only if a short probe says one
    # iteration fits the budget, otherwise the probe-scaled figure is reported and marked as such.
    try:
        torch.set_num_threads(phys)
        g8 = torch.Generator().manual_seed(99)
        x8 = torch.randn(8, 1, M, T, V, C, generator=g8)
        y8 = torch.randint(0, CLASSES, (8, 1), generator=g8)

        def step_n(xx, yy):
            for v in leaves.values():
                v.grad = None
            _, loss = O.recognizer_forward_train(xx, yy, sd, gc['node_type'], gc['edge_type'], plan)
            loss.backward()
        step_n(x8, y8)
        t0 = time.perf_counter()
        step_n(x8, y8)
        t8 = time.perf_counter() - t0
        if 8 * t8 <= 2.5 * budget_s:
            x64 = torch.randn(64, 1, M, T, V, C, generator=g8)
            y64 = torch.randint(0, CLASSES, (64, 1), generator=g8)
            t0 = time.perf_counter()
            step_n(x64, y64)
            t64 = time.perf_counter() - t0
            res['all_cores'] = dict(value=round(64 / t64, 2), unit='clips/s', cores=phys,
                                    sample=f'1 fwd+bwd iteration of the 64-clip batch on all {phys} physical cores, {t64:.1f} s')
        else:
            res['all_cores'] = dict(value=round(8 / t8, 2), unit='clips/s', cores=phys,
                                    sample=f'8-clip probe on all {phys} physical cores ({t8:.1f} s per iteration); the 64-clip '
                                           f'iteration would exceed the time budget, not run')
    except Exception as exc:        # the secondary figure must never take the bench line down
        res['all_cores'] = dict(value=None, error=f'{type(exc).__name__}: {exc}')
    return res


def self_launch(n_gpus, argv):
    """``python bench.py --gpus N`` without a launcher around it: start ``python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <argv>`` as a CHILD process (never exec: a
    process must not be replaced once anything could have touched the GPU — and this one has not: no ``torch.cuda.*`` call
    is made before this point, not even ``is_available()``), relay its stdout (rank 0's JSON line) and return its exit
    code.  The reference's launcher is tools/dist_train.sh:9-11 (``python -m torch.distributed.launch --nproc_per_node``).
    At N = 1 (``DSGCN_BENCH_SELF_LAUNCH=1``: the test of this path on a one-GPU box) the rank builds a 1-rank RCCL group."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL across processes needs it on this driver
    env['DSGCN_BENCH_SELF_LAUNCHED'] = '1'
    if n_gpus == 1:
        env['DSGCN_BENCH_FORCE_DIST'] = '1'
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__), *argv]
    print(f'[bench] self-launch: {" ".join(cmd)}', file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True)
    try:
        for line in child.stdout:
            sys.stdout.write(line)
            sys.stdout.flush()
        return child.wait()
    except BaseException:
        child.kill()           # the exact child we started, nothing by pattern
        child.wait()
        raise


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--clips-per-gpu', type=int, default=CLIPS_PER_GPU)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the BASELINE configs 1 / 3 / 4 / 5 step times')
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying captured hipGraphs')
    ap.add_argument('--cpu-budget', type=float, default=12.0)
    args = ap.parse_args(argv)

    if 'WORLD_SIZE' not in os.environ and (args.gpus > 1 or os.environ.get('DSGCN_BENCH_SELF_LAUNCH') == '1'):
        # the driver's N = 1 form (`python bench.py --gpus N`) asked for N > 1: become the launcher's parent — BEFORE any
        # GPU call — and hand back the children's line and exit code
        raise SystemExit(self_launch(args.gpus, argv))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1:
            raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE=1 is set: unset it (bench.py then launches its own ranks) or '
                             f'launch through python -m torch.distributed.run --nproc-per-node {args.gpus}')
        if rank == 0:
            print(f'[bench] --gpus {args.gpus} but the launcher started {world} ranks: measuring {world}', file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (the hot path has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    force_dist = os.environ.get('DSGCN_BENCH_FORCE_DIST') == '1'      # 1-rank RCCL group: exercises the N>1 code path
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

    import dsgcn_amd
    from dsgcn_amd import native
    native.lib()   # fail loudly if the HIP library is missing

    model = build_model().to(device).train()
    # the step under test is the package's own training engine (ds-gcn_amd/engine.py): graph A (zero-grad + forward +
    # backward + gradient packing) -> RCCL all-reduce of the flat gradient buffer -> graph B (SGD-nesterov update)
    n_eager_warm = min(max(args.warmup, 1), 3) if not args.no_graph else args.warmup
    engine = dsgcn_amd.TrainEngine(model, lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True,
                                   use_graph=not args.no_graph, warmup_eager=n_eager_warm, strict_graph=False,
                                   extra_allreduce=force_dist)
    flat = engine.flat

    B = args.clips_per_gpu
    g = torch.Generator().manual_seed(1234 + rank)
    keypoint = torch.randn(B, 1, M, T, V, C, generator=g).to(device)
    label = torch.randint(0, CLASSES, (B, 1), generator=g).to(device)

    state = {}

    def step():
        state['loss'] = engine.step(keypoint, label)['loss']

    # untimed warm-up: the first steps run eagerly (caches: edge-class lists, allocator pools, pinned tables), then the
    # engine captures the launch-bound inner loop (~800 kernels per step) into its two hipGraphs and replays them
    for _ in range(max(args.warmup, n_eager_warm + 1) if not args.no_graph else args.warmup):
        step()
    torch.cuda.synchronize()
    use_graph = engine.graphed(keypoint, label)
    if not args.no_graph:
        # The graph-vs-eager decision is collective: ranks that replay and ranks that launch eagerly would issue the
        # same collectives, but a silently slower rank makes the aggregate number meaningless — so a failed capture is
        # an error at N > 1 (run with --no-graph to measure the eager path), and only a warning on one GPU.
        ok = torch.tensor([1 if use_graph else 0], device=device, dtype=torch.int32)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            msg = (f'[bench] rank {rank}: hipGraph capture failed on at least one rank '
                   f'({engine.capture_error or "another rank"})')
            if world > 1:
                print(msg + '; aborting (pass --no-graph for an eager run)', file=sys.stderr)
                dist.destroy_process_group()
                raise SystemExit(3)
            print(msg + '; running eagerly', file=sys.stderr)
            engine.use_graph = use_graph = False

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rank_elapsed = [elapsed]
    if world > 1:
        mine = torch.tensor([elapsed], device=device, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_elapsed = [float(t.item()) for t in every]
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    loss_val = float(state['loss'].item())
    assert flat.check_views(), 'a parameter gradient left the flat buffer'
    assert np.isfinite(loss_val), loss_val

    result = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        result = {
            'metric': 'clips/sec fwd+bwd, DS-STGCN NTU-60 3x64x25x2', 'value': round(value, 1), 'unit': 'clips/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'DS-STGCN configs/dsstgcn/ntu60_xsub_3dkp/j.py (DSSTGCN_model.py kwargs), '
                                   f'{B} clips/GPU of 3x{T}x{V}x{M}, 60 classes, train-mode BN, CE loss, '
                                   'fwd+bwd+grad all-reduce+SGD-nesterov per step',
                       'clips_per_gpu': B, 'global_batch': B * world, 'parallelism': f'dp{world}'},
            'final_loss': round(loss_val, 5), 'hip_graph': bool(use_graph),
        }
        if world > 1 or force_dist:
            per = [round(t / args.steps * 1e3, 3) for t in rank_elapsed]
            result['dist'] = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
                              'self_launched': os.environ.get('DSGCN_BENCH_SELF_LAUNCHED') == '1',
                              'rccl_version': '.'.join(str(v) for v in torch.cuda.nccl.version()),
                              'rank_ms_per_step': per, 'rank_ms_min': min(per), 'rank_ms_max': max(per),
                              'grad_bytes': int(flat.flat_g.numel() * flat.flat_g.element_size())}
    # host time per step: how long the Python thread takes to ENQUEUE a step (batch copies, two graph
    # replays, the collective) against the GPU time of the step — the margin the host has before jitter reaches the GPU;
    # at N > 1 this is what decides weak-scaling efficiency, and it is measurable at N = 1
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    for _ in range(10):
        step()
    host_enqueue = (time.perf_counter() - h0) / 10
    torch.cuda.synchronize()
    if rank == 0:
        result['host_enqueue_ms'] = round(host_enqueue * 1e3, 3)
        result['host_enqueue_note'] = ('wall time of the Python thread per engine.step() call with no synchronisation (10 calls '
                                       'after the timed region); the GPU step takes ms_per_step')
    # the optimizer update reported separately (the metric is fwd+bwd; the step above includes the update)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g_b = engine._graphs[(tuple(keypoint.shape), tuple(label.shape))][1] if use_graph else None
    for _ in range(10):
        if use_graph:
            g_b.replay()
        else:
            engine.opt.step()
    e1.record()
    torch.cuda.synchronize()
    if rank == 0:
        result['optimizer_ms'] = round(e0.elapsed_time(e1) / 10, 3)
        result['fwd_bwd_ms'] = round(result['ms_per_step'] - result['optimizer_ms'], 3)
    if world > 1 or force_dist:
        # the gradient exchange by itself (the RCCL all-reduce of the flat buffer between the two graphs), after the timed
        # region: HIP events on the stream the collective is enqueued on, max over ranks
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        for _ in range(10):
            engine._exchange()
        a1.record()
        torch.cuda.synchronize()
        ar = torch.tensor([a0.elapsed_time(a1) / 10], device=device, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(ar, op=dist.ReduceOp.MAX)
        if rank == 0:
            result['dist']['allreduce_ms'] = round(float(ar.item()), 4)
            result['dist']['allreduce_note'] = '10 back-to-back exchanges of the flat gradient buffer, max over ranks'
    if rank == 0:
        # fingerprint of the parameters after the run (W + K steps + the 10 update replays above): two runs of the same
        # command must agree bit for bit, with or without the RCCL group (tests/test_model_gpu.py spawns both)
        import hashlib
        result['param_sha256'] = hashlib.sha256(flat.flat_p.detach().cpu().numpy().tobytes()).hexdigest()[:16]
    if rank == 0 and not args.no_roofline:
        rf = measure_ka_roofline(device, B * M)
        result['roofline'] = rf['k_aggregate_bwd'] | {'kernel': 'k_aggregate_bwd'}
        result['roofline_other'] = ({'k_aggregate_fwd': rf['k_aggregate_fwd']} | measure_kc_roofline(device, B * M) |
                                    measure_kap_roofline(device, B * M))
    if rank == 0:
        result['roofline_step'] = step_roofline(B * M, result['ms_per_step'])
    if rank == 0 and world == 1 and not args.no_other_configs:
        result['other_configs'] = measure_other_configs(device)
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(args.cpu_budget)
        result['cpu_baseline'] = cb
        result['gpu_over_cpu'] = round(result['value'] / cb['value'], 1)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
