"""HBM bytes per kernel per step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over an eager bench run:
    python tools/step_traffic.py FETCH_DIR WRITE_DIR STEPS OUT.csv
FETCH_SIZE is doubled (gfx950 wide-read correction, MI355X_MICROARCH.md, HBM / rocprofv3 section); counters come per
XCD and are summed per dispatch; STEPS = number of steps the profiled command ran (warm-up included)."""
import csv, glob, os, re, sys
from collections import defaultdict


def short(name):
    m = re.search(r'(k_[A-Za-z0-9_]+(?:<[^>]*>)?)', name)
    if m:
        return m.group(1)
    m = re.search(r'(CUDAFunctor\w*<[^>]*>|\w+_kernel\w*|__amd_rocclr_\w+|\w*[Rr]educe\w*|\w*[Cc]at\w*)', name)
    return 'torch:' + (m.group(1) if m else name[:48])


def load(d, counter):
    tot, n = defaultdict(float), defaultdict(set)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f, newline='')):
            if r['Counter_Name'] != counter:
                continue
            k = short(r['Kernel_Name'])
            tot[k] += float(r['Counter_Value'])
            n[k].add(r['Dispatch_Id'])
    return tot, {k: len(v) for k, v in n.items()}


fetch, nf = load(sys.argv[1], 'FETCH_SIZE')
write, _ = load(sys.argv[2], 'WRITE_SIZE')
steps = float(sys.argv[3])
rows = sorted(((k, nf[k] / steps, 2 * fetch[k] / 1024 / steps, write.get(k, 0.0) / 1024 / steps) for k in fetch),
              key=lambda r: -(r[2] + r[3]))
with open(sys.argv[4], 'w', newline='') as fh:
    w = csv.writer(fh)
    w.writerow(['kernel', 'launches_per_step', 'fetch_MB_per_step(FETCH_SIZE x2)', 'write_MB_per_step(WRITE_SIZE)'])
    for k, n, f, wr in rows:
        w.writerow([k, round(n, 1), round(f, 1), round(wr, 1)])
    w.writerow(['TOTAL', round(sum(r[1] for r in rows), 1), round(sum(r[2] for r in rows), 1), round(sum(r[3] for r in rows), 1)])
print('total per step: fetch %.1f GB, write %.1f GB' % (sum(r[2] for r in rows) / 1024, sum(r[3] for r in rows) / 1024))
