"""The dominant-kernel numbers of bench.py's `roofline` object, recomputed from rocprofv3 output:

    python scripts/prof_roofline.py <kernel-trace dir> <steps> [<FETCH_SIZE pmc dir> <WRITE_SIZE pmc dir>]

Dominant kernel = the 3x3 implicit-GEMM forward / data-gradient launches with more than 64 filters (the set bench.py brackets with
HIP events): conv3x3_pp_kernel<...> (conv3x3_tap_kernel<...> in rounds 2-3) and conv_igemm_kernel<bf16, BN=128, .., KS=3, ..>.  Prints their launches per step, average
duration (to be compared with roofline.avg_launch_ms) and, with the two PMC directories, the average HBM bytes per launch
(FETCH_SIZE x 1024 x 2 on gfx950 + WRITE_SIZE x 1024: MI355X_MICROARCH.md "HBM"), which is roofline.traffic."""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import load          # noqa: E402


def dominant(name):
    if 'conv3x3_pp_kernel' in name or 'conv3x3_tap_kernel' in name:       # (round 4: ping-pong tap-fused kernel; rounds 2-3: conv3x3_tap_kernel)
        return True
    if 'conv3x3_s4_kernel' in name:                                          # (round 6: the loader / consumer member)
        return True
    if 'conv_c64_fwd_kernel' in name:        # (round 6: conv2 / conv4 forward with the filter in registers = its 128-filter form <MODE, 4>; the 32-filter form is conv1's data gradient)
        return bool(re.search(r'conv_c64_fwd_kernelILi\d+ELi4E|conv_c64_fwd_kernel<\d+, 4>', name))
    m = re.search(r'conv_igemm_kernelI(DF16b|f)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E', name)      # T, BN, WGN, NSTAGE, KS
    if m:
        return int(m.group(2)) == 128 and int(m.group(5)) == 3
    m = re.search(r'conv_igemm_kernel<[^,]+, (\d+), \d+, \d+, (\d+)', name)                   # demangled form
    return bool(m) and int(m.group(1)) == 128 and int(m.group(2)) == 3


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    rows = load(path)
    adam = [i for i, r in enumerate(rows) if ('adam_kernel' in r[0] or 'adam_filter_prep_kernel' in r[0])]
    sel = rows[adam[-steps - 1] + 1: adam[-1] + 1]
    dom = [(n, e - s) for n, s, e in sel if dominant(n)]
    tot = sum(d for _, d in dom)
    print('# dominant kernel (3x3 implicit GEMM, > 64 filters) in the last %d training steps of the trace' % steps)
    print('launches per step: %.1f; average duration %.2f us; %.1f us per step (%.1f %% of the summed kernel time)'
          % (len(dom) / steps, tot / len(dom) / 1e3, tot / steps / 1e3, 100.0 * tot / sum(e - s for _, s, e in sel)))
    by = {}
    for n, d in dom:
        k = 'conv3x3_pp_kernel' if 'conv3x3_pp' in n else 'conv_c64_fwd_kernel' if 'conv_c64' in n else 'conv3x3_s4_kernel' if 'conv3x3_s4' in n else 'conv3x3_tap_kernel' if 'conv3x3_tap' in n else re.sub(r'EEv.*', '', n.replace('_Z17', ''))[:70]
        a = by.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += d
    for k, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print('  %-72s %5.1f launches/step  avg %7.2f us' % (k, c / steps, d / c / 1e3))
    if len(sys.argv) >= 5:
        import csv, glob
        csv.field_size_limit(1 << 30)

        def counter(p, cname):
            rows = {}
            for f in glob.glob(os.path.join(p, '**', '*counter_collection.csv'), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r['Counter_Name'] == cname:
                        e = rows.setdefault(int(r['Dispatch_Id']), [r['Kernel_Name'], 0.0])
                        e[1] += float(r['Counter_Value'])
            rows = [rows[k] for k in sorted(rows)]
            ad = [i for i, r in enumerate(rows) if ('adam_kernel' in r[0] or 'adam_filter_prep_kernel' in r[0])]
            return rows[ad[-2] + 1: ad[-1] + 1]
        fe, wr = counter(sys.argv[3], 'FETCH_SIZE'), counter(sys.argv[4], 'WRITE_SIZE')
        f = [v * 1024 * 2 for n, v in fe if dominant(n)]
        w = [v * 1024 for n, v in wr if dominant(n)]
        print('HBM traffic of those launches in the last step of the PMC passes: %d launches, fetch %.1f MB + write %.1f MB = %.1f MB per launch'
              % (len(f), sum(f) / len(f) / 1e6, sum(w) / len(w) / 1e6, (sum(f) / len(f) + sum(w) / len(w)) / 1e6))


main()
