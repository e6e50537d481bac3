"""f32 MFMA issue-rate probe (lab library: csrc/lab/diag.hip): sustained TFLOP/s of v_mfma_f32_32x32x2_f32 chains."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsgcn_amd import native
lib = native.lab_lib(); st = torch.cuda.current_stream().cuda_stream
out = torch.empty(4096 * 256, device='cuda')
for blocks in (256, 512, 1024):
    for nacc in (1, 2, 4):
        iters = 20000
        lib.dsgcn_diag_mfma_probe(out.data_ptr(), blocks, 100, nacc, st); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.dsgcn_diag_mfma_probe(out.data_ptr(), blocks, iters, nacc, st); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        n_mfma = blocks * 4 * iters * nacc
        print(f'blocks={blocks} nacc={nacc}: {ms:.2f} ms, {n_mfma*4096/ms/1e9:.1f} TFLOP/s, cycles/MFMA/SIMD @2.4GHz = {ms*1e-3*2.4e9/(n_mfma/1024):.1f}')
