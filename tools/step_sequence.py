"""The kernel sequence of one replayed step from a rocprofv3 --kernel-trace CSV (steps delimited by k_pack): name and
duration in launch order, runs of small kernels marked.     python tools/step_sequence.py <kernel_trace.csv> [out.txt]"""
import csv, re, sys
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1])))
packs = [i for i, r in enumerate(rows) if 'k_pack' in r[2]]
a, b = packs[-3], packs[-2]
def short(n):
    m = re.search(r'(k_[A-Za-z0-9_]+(?:<[^>]*>)?)', n)
    return m.group(1) if m else 'torch:' + re.sub(r'\W+', '_', n)[:40]
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
small = 0.0; runs = {}
prev_small = None
for s, e, n in rows[a + 1:b + 1]:
    d = (e - s) / 1e3
    out.write(f'{d:8.1f} us  {short(n)}\n')
    if d < 8:
        small += d
        if prev_small is not None:
            key = (prev_small, short(n)); runs[key] = runs.get(key, 0) + 1
        prev_small = short(n)
    else:
        prev_small = None
out.write(f'# kernels < 8 us: {small:.0f} us in total; adjacent pairs of small kernels:\n')
for k, v in sorted(runs.items(), key=lambda kv: -kv[1])[:20]:
    out.write(f'#   {v:3d} x {k[0]} -> {k[1]}\n')
