"""Per-launch HBM bytes of the InfoNCE / graph kernels from the PMC passes of tools/pmc_hbm.sh.  rocprofv3 reports
FETCH_SIZE / WRITE_SIZE in KiB; gfx950 counts 64 B per 128-B request on wide streaming reads, so FETCH_SIZE is doubled
(MI355X_MICROARCH.md, HBM section; re-verified on EMA/SGD in profiles/r02_pmc_traffic.json)."""
import collections, csv, json, re, sys
root = sys.argv[1]
ALG = {'moco_logits_persist_kernel<4>': None}


def short(name):
    s = re.sub(r'\(anonymous namespace\)::', '', name)
    return re.sub(r'^void ', '', s).split('(')[0]


def load(path, counter):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    with open(path) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] == counter:
                k = short(r['Kernel_Name'])
                if 'moco_logits' in k or 'tmix' in k:
                    e = agg[k][int(r['Grid_Size'])]
                    e[0] += 1; e[1] += float(r['Counter_Value']) * 1024.0
    return agg


f = load(root + '/fetch/f_counter_collection.csv', 'FETCH_SIZE')
w = load(root + '/write/w_counter_collection.csv', 'WRITE_SIZE')
out = {}
for k in f:
    for grid, (n, fb) in f[k].items():
        wn, wb = w.get(k, {}).get(grid, [0, 0.0])
        out['%s grid=%d' % (k, grid)] = dict(launches=n, fetch_bytes_per_launch=round(2 * fb / n), write_bytes_per_launch=round(wb / max(wn, 1)),
                                             hbm_bytes_per_launch=round(2 * fb / n + wb / max(wn, 1)))
print(json.dumps(out, indent=1))
