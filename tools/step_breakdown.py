"""Per-step kernel breakdown from a rocprofv3 rocpd database (--kernel-trace): isolates the graph-replayed
training steps (delimited by the SGD kernel) and prints time per kernel family for one steady-state step."""
import collections
import re
import sqlite3
import sys

if sys.argv[1].endswith('.csv'):          # rocprofv3 --kernel-trace --output-format csv
    import csv
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r['VGPR_Count']),
                         int(r['Accum_VGPR_Count']), int(r['LDS_Block_Size'])))
    rows.sort(key=lambda r: r[1])
else:                                     # rocpd database (default output of rocprofv3 on ROCm 7.2)
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, start, end, vgpr_count, accum_vgpr_count, lds_size from kernels order by start"))
sgd = [i for i, r in enumerate(rows) if '::sgd_kernel' in r[0]]
steps = []
for a, b in zip(sgd[:-1], sgd[1:]):
    steps.append((rows[b][2] - rows[a][2], a + 1, b + 1))
# steady state = most common dispatch count
cnt = collections.Counter(b - a for _, a, b in steps).most_common(1)[0][0]
steady = [s for s in steps if s[2] - s[1] == cnt]
steady = steady[-int(sys.argv[2]) if len(sys.argv) > 2 else -10:]
agg = collections.defaultdict(lambda: [0, 0.0])
regs = {}
wall = 0.0
for dur, a, b in steady:
    wall += dur
    for name, st, en, vg, ag, lds in rows[a:b]:
        short = re.sub(r'\(anonymous namespace\)::', '', name)
        short = re.sub(r'^void ', '', short).split('(')[0]
        agg[short][0] += 1
        agg[short][1] += en - st
        regs[short] = (vg, ag, lds)
n = len(steady)
busy = sum(v[1] for v in agg.values())
print('steady steps: %d  dispatches/step: %d  wall/step %.3f ms  kernel-busy/step %.3f ms' % (n, cnt, wall / n / 1e6, busy / n / 1e6))
print('%-56s %7s %10s %7s %9s  vgpr agpr lds' % ('kernel', 'calls', 'ms/step', '%', 'avg_us'))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-56s %7.1f %10.4f %7.2f %9.2f  %s' % (k[:56], c / n, t / n / 1e6, 100 * t / busy, t / c / 1e3, regs[k]))
