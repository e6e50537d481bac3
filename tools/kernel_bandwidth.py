"""Achieved HBM bandwidth per kernel of the replayed step: HBM bytes (rocprofv3 --pmc passes over an eager step,
tools/step_traffic.py: FETCH_SIZE x2 + WRITE_SIZE, MiB) divided by the kernel's time in the graph replay
(tools/step_sequence.py).  The practical ceiling on this part is ~4.5-5.5 TB/s (a torch.copy_ of the same planes: 4.9);
kernels at it can only get cheaper by moving fewer bytes, kernels far below it are bound by something else.
    python tools/kernel_bandwidth.py profiles/r06/step_hbm_traffic.csv profiles/r06/step_sequence.txt"""
import collections, csv, re, sys

tr = {r['kernel']: (float(r['fetch_MB_per_step(FETCH_SIZE x2)']) + float(r['write_MB_per_step(WRITE_SIZE)'])) * 1.048576
      for r in csv.DictReader(open(sys.argv[1])) if r['kernel'] != 'TOTAL'}
t, c = collections.defaultdict(float), collections.Counter()
for line in open(sys.argv[2]):
    m = re.match(r'\s*([\d.]+) us\s+(.*)$', line)
    if m:
        t[m.group(2).strip()] += float(m.group(1))
        c[m.group(2).strip()] += 1
rows = sorted(((t[k], k, c[k], mb) for k, mb in tr.items() if k in t), reverse=True)
print(f"{'us/step':>8} {'calls':>5} {'MB/step':>8} {'TB/s':>5}  kernel")
tt = tm = 0.0
for us, k, n, mb in rows:
    print(f'{us:8.0f} {n:5d} {mb:8.0f} {mb / us:5.2f}  {k}')
    tt += us
    tm += mb
print(f'{tt:8.0f} {sum(c[k] for _, k, _, _ in rows):5d} {tm:8.0f} {tm / tt:5.2f}  all kernels with counters')
print(f'# step: {sum(t.values()):.0f} us in {sum(c.values())} kernels')
