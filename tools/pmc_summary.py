"""Compact per-kernel summary of rocprofv3 --pmc output directories.

    python tools/pmc_summary.py OUT.csv DIR [DIR ...]

Every DIR holds one counter-collection pass (``*_counter_collection.csv``).  Rows are averaged per (kernel, grid
size) over the dispatches of each pass; only this library's kernels (``k_*``) are kept, template arguments included,
argument lists dropped.  Columns: kernel, grid, wg, vgpr, lds, dispatches, avg_us, then one column per counter."""
import csv
import glob
import os
import re
import sys
from collections import OrderedDict, defaultdict


def short(name):
    m = re.search(r'(k_[A-Za-z0-9_]+(?:<[^>]*>)?)', name)
    return m.group(1) if m else None


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    rows = OrderedDict()
    counters = []
    for d in dirs:
        for f in sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)):
            per = defaultdict(lambda: defaultdict(list))
            meta = {}
            with open(f, newline='') as fh:
                for r in csv.DictReader(fh):
                    k = short(r['Kernel_Name'])
                    if k is None:
                        continue
                    key = (k, int(r['Grid_Size']))
                    per[key][r['Counter_Name']].append((r['Dispatch_Id'], float(r['Counter_Value'])))
                    per[key]['__us'].append((r['Dispatch_Id'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
                    meta[key] = (r['Workgroup_Size'], r['VGPR_Count'], r['Accum_VGPR_Count'], r['LDS_Block_Size'])
            for key, cs in per.items():
                row = rows.setdefault(key, OrderedDict(kernel=key[0], grid=key[1], wg=meta[key][0], vgpr=meta[key][1],
                                                       agpr=meta[key][2], lds=meta[key][3]))
                for c, vals in cs.items():
                    by_disp = defaultdict(float)
                    for disp, v in vals:
                        by_disp[disp] = v if c == '__us' else by_disp[disp] + v      # counters come per XCD/SE: sum
                    avg = sum(by_disp.values()) / len(by_disp)
                    if c == '__us':
                        row.setdefault('dispatches', len(by_disp))
                        row.setdefault('avg_us', round(avg, 2))
                    else:
                        row[c] = round(avg, 1)
                        if c not in counters:
                            counters.append(c)
    cols = ['kernel', 'grid', 'wg', 'vgpr', 'agpr', 'lds', 'dispatches', 'avg_us'] + counters
    with open(out, 'w', newline='') as fh:
        w = csv.DictWriter(fh, fieldnames=cols)
        w.writeheader()
        for row in rows.values():
            w.writerow({c: row.get(c, '') for c in cols})
    print(f'{out}: {len(rows)} kernels, counters: {" ".join(counters)}')


if __name__ == '__main__':
    main()
