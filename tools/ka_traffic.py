"""Turn the two rocprofv3 --pmc passes over tools/ka_once.py into profiles/rNN/ka_traffic.json:
    python tools/ka_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
HBM bytes per launch = (FETCH_SIZE[KB] x 2 + WRITE_SIZE[KB]) x 1024, averaged over the 10-layer mix (the gfx950
FETCH_SIZE counter reports half of wide reads: MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv, json, sys

def per_kernel(path, counter):
    tot, cnt = {}, {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = r['Kernel_Name']
        key = 'k_aggregate_fwd' if 'k_aggregate_fwd' in name else ('k_aggregate_bwd' if 'k_aggregate_bwd' in name else None)
        if key is None:
            continue
        tot[key] = tot.get(key, 0.0) + float(r['Counter_Value'])
        cnt[key] = cnt.get(key, 0) + 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt

fetch, nf = per_kernel(sys.argv[1], 'FETCH_SIZE')
write, nw = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in fetch:
    out[k] = {'FETCH_SIZE_KB_per_launch': fetch[k], 'WRITE_SIZE_KB_per_launch': write[k], 'launches_profiled': nf[k],
              'hbm_bytes_per_launch': int((2 * fetch[k] + write[k]) * 1024)}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out, indent=1))
