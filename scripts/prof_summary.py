"""Summarises a rocprofv3 --kernel-trace run (rocpd SQLite or CSV) into a per-kernel table (markdown)."""
import csv, glob, os, sqlite3, sys

def load(path):
    rows = []
    dbs = glob.glob(os.path.join(path, '**', '*.db'), recursive=True)
    if dbs:
        cur = sqlite3.connect(dbs[0]).cursor()
        for name, s, e in cur.execute('select name, start, end from kernels order by start'):
            rows.append((name, s, e))
        return rows
    for f in glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])))
    rows.sort(key=lambda r: r[1])
    return rows

def short(n):
    n = n.replace('void ', '')
    for a, b in (('_Z17conv_igemm_kernel', 'conv_igemm_kernel'), ('_Z17conv_wgrad_kernel', 'conv_wgrad_kernel')):
        n = n.replace(a, b)
    return n[:96]

def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    rows = load(path)
    adam = [i for i, r in enumerate(rows) if ('adam_kernel' in r[0] or 'adam_filter_prep_kernel' in r[0])]
    lo, hi = adam[-steps - 1] + 1, adam[-1] + 1          # the last `steps` training steps
    sel = rows[lo:hi]
    wall = (sel[-1][2] - sel[0][1]) / 1e6 / steps
    agg = {}
    for n, s, e in sel:
        a = agg.setdefault(short(n), [0, 0.0])
        a[0] += 1; a[1] += (e - s) / 1e3
    tot = sum(v[1] for v in agg.values())
    print('# rocprofv3 --kernel-trace summary: last %d training steps of bench.py' % steps)
    print('wall per step (first kernel start -> last kernel end): %.3f ms; sum of kernel durations per step: %.3f ms; launches per step: %d\n'
          % (wall, tot / 1e3 / steps, len(sel) // steps))
    print('| kernel | calls/step | avg us | us/step | % of kernel time |')
    print('|---|---:|---:|---:|---:|')
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('| `%s` | %.1f | %.1f | %.1f | %.2f |' % (n, c / steps, t / c, t / steps, 100 * t / tot))

if __name__ == "__main__":
    main()
