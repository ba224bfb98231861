"""Per-step view of a rocprofv3 kernel trace: picks one graph-replayed step (between two clip_to_ndhwc
launches) and prints every dispatch with duration, gap to the previous one, grid and LDS size."""
import csv, sys, collections
path = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'clip_to_ndhwc' in r['Kernel_Name']]
a, b = starts[which], starts[which + 1]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp']); prev_end = t0
busy = 0; agg = collections.OrderedDict()
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    d = (e - s) / 1e3; gap = (s - prev_end) / 1e3; prev_end = max(prev_end, e); busy += d
    name = r['Kernel_Name'].replace('void ', '').replace('trace::', '')[:60]
    if '-v' in sys.argv:
        print('%8.1f %7.1f us gap %6.1f  grid %7s x%s x%s lds %6s vgpr %s  %s' % ((s - t0) / 1e3, d, gap, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y'], r['Grid_Size_Z'], r['LDS_Block_Size'], r['VGPR_Count'], name))
    k = name
    agg.setdefault(k, [0, 0.0]); agg[k][0] += 1; agg[k][1] += d
span = (prev_end - t0) / 1e3
print('step %d: %d dispatches, span %.1f us, busy %.1f us' % (which, len(step), span, busy))
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print('%5d %8.1f us  %5.1f%%  %s' % (n, d, 100 * d / span, k))
