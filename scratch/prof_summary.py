"""Summarise a rocprofv3 results.db: per-kernel GPU time of the LAST bench step."""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,grid_x,workgroup_x from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if 'clip_to_ndhwc' in r[0]]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
a = idx[-nsteps]
step = rows[a:]
print(len(step) / nsteps, 'kernels/step; span ms', (step[-1][2] - step[0][1]) / 1e6 / nsteps, 'busy ms', sum(r[2] - r[1] for r in step) / 1e6 / nsteps)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    n = r[0].replace('void ', '').replace('trace::', '')
    n = n.split('(')[0][:70]
    agg[n][0] += 1
    agg[n][1] += (r[2] - r[1]) / 1e3
for n, (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%8.1f us %5.1f x  %s' % (us / nsteps, cnt / nsteps, n))
