"""Per-launch listing (start offset, duration, gap to the previous kernel) of the last training step of a rocprofv3 kernel trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import load, short
rows = load(sys.argv[1])
adam = [i for i, r in enumerate(rows) if ('adam_kernel' in r[0] or 'adam_filter_prep_kernel' in r[0])]
sel = rows[adam[-2] + 1: adam[-1] + 1]
t0 = sel[0][1]
prev_end = t0
print('# last training step: %d launches, %.3f ms' % (len(sel), (sel[-1][2] - t0) / 1e6))
print('%8s %8s %7s  kernel' % ('t_us', 'dur_us', 'gap_us'))
for n, s, e in sel:
    print('%8.1f %8.1f %7.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short(n)[:110]))
    prev_end = max(prev_end, e)
