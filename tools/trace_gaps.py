"""Busy time vs wall time of the replayed step from a rocprofv3 --kernel-trace CSV:
    python tools/trace_gaps.py <kernel_trace.csv>
Steps are delimited by k_pack (one per step); prints wall, union-busy and idle time of the last few steps and the idle
time attributed to the kernel that FOLLOWS each gap (top 15)."""
import csv, sys, re
from collections import defaultdict
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
packs = [i for i, r in enumerate(rows) if 'k_pack' in r[2]]
def short(n):
    m = re.search(r'(k_[A-Za-z0-9_]+)', n)
    return m.group(1) if m else n[:40]
for a, b in list(zip(packs[:-1], packs[1:]))[-3:]:
    seg = rows[a + 1:b + 1]
    wall = seg[-1][1] - seg[0][0]
    busy = 0; cur_e = seg[0][0]; gaps = defaultdict(float); ngap = defaultdict(int)
    for s, e, n in seg:
        if s > cur_e:
            gaps[short(n)] += s - cur_e; ngap[short(n)] += 1
        if e > cur_e:
            busy += e - max(s, cur_e); cur_e = e
    tot = sum(e - s for s, e, n in seg)
    print(f'step: {len(seg)} kernels, wall {wall/1e6:.3f} ms, busy {busy/1e6:.3f} ms, idle {(wall-busy)/1e6:.3f} ms, sum of durations {tot/1e6:.3f} ms')
a, b = packs[-3], packs[-2]          # a steady-state step: list its large gaps with both neighbours
seg = rows[a + 1:b + 1]
cur_e = seg[0][0]; prev = ''
for s_, e_, n_ in seg:
    if s_ - cur_e > 8000:
        print(f'  gap {(s_-cur_e)/1e3:7.1f} us  after {short(prev):28s} before {short(n_)}')
    if e_ > cur_e:
        cur_e = e_; prev = n_
# kernels running concurrently with K-B (the side-stream branch)
for s_, e_, n_ in seg:
    if 'k_dynadj' in n_:
        ov = [(short(n2), min(e_, e2) - max(s_, s2)) for s2, e2, n2 in seg if n2 is not n_ and min(e_, e2) > max(s_, s2)]
        print(f'  {short(n_)} {(e_-s_)/1e3:.1f} us overlaps: ' + ', '.join(f'{k} {v/1e3:.1f}' for k, v in ov[:6]))
top = sorted(gaps.items(), key=lambda kv: -kv[1])[:15]
for k, v in top:
    print(f'  idle before {k:32s} {v/1e3:8.1f} us over {ngap[k]} gaps ({v/1e3/max(ngap[k],1):.2f} us each)')
