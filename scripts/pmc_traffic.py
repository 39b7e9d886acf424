"""HBM traffic per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md "HBM"):
FETCH_SIZE is reported in KiB-like units of 1024 B and, on gfx950, counts 128-B requests at 64 B -> doubled here.
WRITE_SIZE is left uncorrected and calibrated against kernels with a known byte count (adam_kernel, bn_leaky_kernel).

    python scripts/pmc_traffic.py <dir of --pmc FETCH_SIZE run> <dir of --pmc WRITE_SIZE run>
Prints the per-dispatch table of the last training step (dispatch order = program order: PMC serialises kernels)."""
import csv, glob, os, sys
csv.field_size_limit(1 << 30)

def load(path, counter):
    rows = {}
    for f in glob.glob(os.path.join(path, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            d = int(r['Dispatch_Id'])
            e = rows.setdefault(d, [r['Kernel_Name'], int(r.get('Grid_Size', 0) or 0), int(r.get('Workgroup_Size', 0) or 0), 0.0])
            e[3] += float(r['Counter_Value'])
    return [rows[k] for k in sorted(rows)]

def short(n):
    n = n.replace('void ', '')
    return n[:84]

def last_step(rows):
    adam = [i for i, r in enumerate(rows) if ('adam_kernel' in r[0] or 'adam_filter_prep_kernel' in r[0])]
    return rows[adam[-2] + 1: adam[-1] + 1]

def main():
    fe = last_step(load(sys.argv[1], 'FETCH_SIZE'))
    wr = last_step(load(sys.argv[2], 'WRITE_SIZE'))
    assert len(fe) == len(wr), (len(fe), len(wr))
    print('# HBM traffic per launch, last training step (bf16, batch 16, 416x416); bytes = counter x 1024; FETCH doubled (gfx950)\n')
    print('| # | kernel | blocks | fetch MB (x2 corrected) | write MB (raw) | phase |')
    print('|---:|---|---:|---:|---:|---|')
    phase = 'fwd'
    agg = {}
    for i, (a, b) in enumerate(zip(fe, wr)):
        assert a[0] == b[0], (a[0], b[0])
        name = short(a[0])
        if 'loss_kernel' in name:
            phase = 'bwd'
        f_mb = a[3] * 1024 * 2 / 1e6
        w_mb = b[3] * 1024 / 1e6
        blocks = a[1] // max(a[2], 1)
        print('| %d | %s | %d | %.2f | %.2f | %s |' % (i, name, blocks, f_mb, w_mb, phase))
        g = agg.setdefault((name.split('(')[0][:60], phase), [0, 0.0, 0.0])
        g[0] += 1; g[1] += f_mb; g[2] += w_mb
    print('\n## per kernel and phase\n\n| kernel | phase | launches | avg fetch MB | avg write MB | total MB |\n|---|---|---:|---:|---:|---:|')
    for (n, ph), g in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print('| %s | %s | %d | %.2f | %.2f | %.1f |' % (n, ph, g[0], g[1] / g[0], g[2] / g[0], g[1] + g[2]))
    print('\nstep total: fetch %.1f MB, write %.1f MB' % (sum(g[1] for g in agg.values()), sum(g[2] for g in agg.values())))

main()
