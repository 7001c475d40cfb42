"""Aggregate rocprofv3 counter CSVs of tools/pmc_step.sh per kernel family over the steady-state training
steps (delimited by the SGD kernel): launches, FETCH_SIZE and WRITE_SIZE per step and per launch.
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3.  gfx950 correction (MI355X_MICROARCH.md, HBM section):
FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) streaming reads -> doubled for the kernels whose
loads are float4 (flag 'x2'); narrower accesses are uncalibrated there, so they are calibrated HERE on kernels
of this very run whose byte counts are known exactly (EMA / SGD over the flat parameter arena)."""
import collections
import csv
import json
import re
import sys

root = sys.argv[1]
ARENA = float(sys.argv[2]) if len(sys.argv) > 2 else 33.51e6     # parameter elements (R(2+1)D-18 + head)


def short(name):
    s = re.sub(r'\(anonymous namespace\)::', '', name)
    s = re.sub(r'^void ', '', s).split('(')[0]
    return s


def load(path, counter):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] == counter:
                rows.append((int(r['Start_Timestamp']), short(r['Kernel_Name']), float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    rows.sort()
    return rows


def per_step(rows):
    idx = [i for i, r in enumerate(rows) if r[1] == 'sgd_kernel']
    spans = [(a + 1, b + 1) for a, b in zip(idx[:-1], idx[1:])]
    cnt = collections.Counter(b - a for a, b in spans).most_common(1)[0][0]
    spans = [s for s in spans if s[1] - s[0] == cnt][-4:]
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for a, b in spans:
        for _, name, val, dur in rows[a:b]:
            e = agg[name]
            e[0] += 1; e[1] += val; e[2] += dur
    n = len(spans)
    return {k: (v[0] / n, v[1] / n * 1024.0, v[2] / n) for k, v in agg.items()}, n


fetch, nf = per_step(load(root + '/fetch/f_counter_collection.csv', 'FETCH_SIZE'))
write, nw = per_step(load(root + '/write/w_counter_collection.csv', 'WRITE_SIZE'))
out = {'steps_averaged': [nf, nw], 'unit': 'bytes per training step (raw counter, KiB*1024)', 'kernels': {}}
for k in sorted(fetch, key=lambda k: -(fetch[k][1] + write.get(k, (0, 0, 0))[1])):
    f, w = fetch[k], write.get(k, (0, 0.0, 0))
    out['kernels'][k] = dict(launches=f[0], fetch_raw=round(f[1]), write_raw=round(w[1]), ns_under_pmc=round(f[2]))
cal = {}
if 'ema_kernel' in fetch:
    cal['ema_kernel'] = dict(known_read=2 * 4 * ARENA, known_write=4 * ARENA, fetch_raw=fetch['ema_kernel'][1], write_raw=write['ema_kernel'][1],
                             fetch_ratio=fetch['ema_kernel'][1] / (2 * 4 * ARENA), write_ratio=write['ema_kernel'][1] / (4 * ARENA))
if 'sgd_kernel' in fetch:
    cal['sgd_kernel'] = dict(known_read=3 * 4 * ARENA, known_write=2 * 4 * ARENA, fetch_raw=fetch['sgd_kernel'][1], write_raw=write['sgd_kernel'][1],
                             fetch_ratio=fetch['sgd_kernel'][1] / (3 * 4 * ARENA), write_ratio=write['sgd_kernel'][1] / (2 * 4 * ARENA))
out['calibration'] = cal
json.dump(out, sys.stdout, indent=1)
