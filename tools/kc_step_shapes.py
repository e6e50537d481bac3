"""Which 1x1 convs the training step runs and what each costs in place: one eager forward + backward of the bench model
with the K-C entry points wrapped (HIP events around every call, a synchronize after it), aggregated by shape and
operand mode.  Cold-operand times (the step's own producer / consumer order), unlike tools/kc_bench.py's repeated
launches on one buffer set."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dsgcn_amd import native

lib = native.lib()
POS = {'dsgcn_pwconv_fwd': 12, 'dsgcn_pwconv_dgrad': 17, 'dsgcn_pwconv_wgrad': 16, 'dsgcn_pwconv_bwd': 18}
rec = collections.OrderedDict()


def wrap(name, npos):
    fn = getattr(lib, name)

    def call(*a):
        n, Ci, Co, T, V = a[npos:npos + 5]
        stride = a[npos + 5] if name != 'dsgcn_pwconv_bwd' else 1
        aug = a[npos + 6] if name != 'dsgcn_pwconv_bwd' else 0
        if name == 'dsgcn_pwconv_fwd':
            mode = ('2' if a[3] else '') + ('a' if a[1] else '') + ('r' if a[6] else '')
        else:
            A0 = a[12] if name == 'dsgcn_pwconv_dgrad' else (a[11] if name == 'dsgcn_pwconv_wgrad' else a[10])
            mode = ('2' if a[3] else '') + ('a' if a[1] else '') + ('r' if a[6] else '') + ('+bn' if A0 else '')
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*a)
        e1.record()
        torch.cuda.synchronize()
        key = (name[13:], n, Ci, Co, T, V, stride, aug, mode or 'plain')
        rec.setdefault(key, []).append(e0.elapsed_time(e1) * 1e3)
        return rc
    setattr(lib, name, call)


def main():
    dev = torch.device('cuda:0')
    model = bench.build_model().to(dev).train()
    x = torch.randn(64, 1, 2, 64, 25, 3, device=dev)
    y = torch.randint(0, 60, (64, 1), device=dev)

    def step():
        model.zero_grad(set_to_none=True)
        out = model.train_step(dict(keypoint=x, label=y), None)
        out['loss'].backward()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    for name, npos in POS.items():
        wrap(name, npos)
    step()
    tot = collections.defaultdict(float)
    print(f'{"call":6s} {"n":>4s} {"Ci":>4s} {"Co":>4s} {"T":>3s} {"V":>3s} s a {"mode":8s} {"calls":>5s} {"us avg":>8s} {"us/step":>8s}')
    for k, v in sorted(rec.items(), key=lambda kv: -sum(kv[1])):
        print(f'{k[0]:6s} {k[1]:4d} {k[2]:4d} {k[3]:4d} {k[4]:3d} {k[5]:3d} {k[6]} {k[7]} {k[8]:8s} {len(v):5d} {sum(v) / len(v):8.1f} {sum(v):8.1f}')
        tot[k[0]] += sum(v)
    print({k: round(v) for k, v in tot.items()})


if __name__ == '__main__':
    main()
