"""RNG-free weights and inputs of the full-size fixtures (tests/golden/full_size.npz): shared by the generator
(reference side) and the tests (this repo's side), so only outputs need to be stored."""
import torch


def closed_form_fill(module):
    """Deterministic, RNG-free weights for the full-size fixtures: every floating tensor gets
    scale * sin(0.37*i + phase(name)); BatchNorm scales ~1, running_var > 0; alpha/beta/add_coeff alive."""
    import zlib
    with torch.no_grad():
        for k, v in module.state_dict().items():
            if not v.dtype.is_floating_point:
                continue
            i = torch.arange(v.numel(), dtype=torch.float64)
            phase = (zlib.crc32(k.encode()) % 1000) / 1000.0 * 6.283185307179586
            wave = torch.sin(0.37 * i + phase).reshape(v.shape)
            if k.endswith('running_var'):
                t = 1.0 + 0.25 * wave
            elif k.endswith('running_mean'):
                t = 0.05 * wave
            elif k.endswith('weight') and v.dim() == 1:            # BatchNorm gamma
                t = 1.0 + 0.1 * wave
            elif k.endswith('bias'):
                t = 0.02 * wave
            elif k.endswith(('alpha', 'beta', 'add_coeff')):
                t = 0.5 * wave
            elif k.endswith('.A') or k == 'backbone.A':
                t = 0.04 + 0.02 * wave
            else:
                fan_in = max(1, v[0].numel()) if v.dim() > 1 else 1
                t = wave * (1.5 / fan_in) ** 0.5
            v.copy_(t.to(v.dtype))


def counter_input(N, T, V, classes):
    i = torch.arange(N * 2 * T * V * 3, dtype=torch.float64)
    x = (torch.sin(0.0137 * i) + 0.3 * torch.cos(0.00071 * i * i % 6.283185307179586)).reshape(N, 1, 2, T, V, 3).float()
    y = (torch.arange(N) * 7 % classes).reshape(N, 1)
    return x, y
