"""Per-dispatch SQ counters of scripts/one_layer.py (several rocprofv3 --pmc passes) -> one row per kernel launch of the last repeat."""
import csv, glob, os, sys
csv.field_size_limit(1 << 30)
root = sys.argv[1]
data = {}     # dispatch order index -> {name, grid, counters}
for d in sorted(glob.glob(os.path.join(root, 'p*'))):
    if not os.path.isdir(d):
        continue
    rows = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = int(r['Dispatch_Id'])
            e = rows.setdefault(k, {'name': r['Kernel_Name'], 'grid': int(r['Grid_Size']) // max(int(r['Workgroup_Size']), 1)})
            e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    conv = [rows[k] for k in sorted(rows) if any(t in rows[k]['name'] for t in ('conv_igemm', 'conv_wgrad', 'conv3x3_tap', 'conv3x3_pp', 'mfma_loop', 'conv_c64', 'conv_c32'))]      # (round 6: conv2's forward is conv_c64_fwd_kernel)
    for i, e in enumerate(conv):
        data.setdefault(i, {}).update(e)
# MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (MFMA_NORM x GRBM_GUI_ACTIVE).  Round 1 divided by SQ_BUSY_CYCLES and printed > 100 %: the two
# counters are summed over different unit counts.  MFMA_NORM is calibrated, not assumed: scripts/experiments/mfma_peak.hip (a pure
# back-to-back MFMA loop, 16 waves per CU = every SIMD's matrix pipe always busy) must read ~100 %: GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES (= 32 x SQ_INSTS_MFMA) over all 1024 SIMDs -> 128.
MFMA_NORM = 128.0      # SIMDs per XCD; the calibration run (argv[2], informational) reads %s at 16 waves per CU
CAL = sys.argv[2] if len(sys.argv) > 2 else 'n/a'
keys = ['SQ_WAVES', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_LDS_BANK_CONFLICT',
        'SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'SQ_INSTS_VMEM_RD', 'SQ_LDS_IDX_ACTIVE', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_LDS',
        'GRBM_GUI_ACTIVE', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_MISC', 'SQ_ACTIVE_INST_VMEM']
print('# SQ counters per launch (scripts/one_layer.py: fwd igemm + wgrad of 6 layers, 3 repeats; last repeat shown)\n')
print('MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE); calibration (pure MFMA loop, 16 waves per CU): ratio %s of 128\n' % CAL)
print('| kernel | blocks | wave-cycles M | wait_any % | wait_inst % | active_inst % | MFMA pipe busy % | LDS idx active/busy % | bank conf/LDS active % | VALU/MFMA | SALU/MFMA | LDS/MFMA | VMEM/MFMA | wait_inst_lds % |')
print('|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|')
n = len(data)
for i in [j for j in range(n) if j % 6 >= 4]:        # per layer: 3 x (igemm, wgrad); the last repeat
    e = data[i]
    g = lambda k: e.get(k, 0.0)
    wc = g('SQ_WAVE_CYCLES') or 1.0
    mf = g('SQ_INSTS_MFMA') or 1.0
    nm = e['name'].replace('void ', '')
    nm = ('ping-pong ' + nm[nm.find('kernelI') + 7: nm.find('EEv')][:30] if 'conv3x3_pp' in nm else 'tap-fused ' if 'conv3x3_tap' in nm else 'mfma_loop ' if 'mfma_loop' in nm else 'igemm ' if 'igemm' in nm else 'filter-in-registers conv_c64 ' if 'conv_c64' in nm else 'conv_c32 ' if 'conv_c32' in nm else 'wgrad ') + nm[nm.find('IDF16b') + 6: nm.find('EEv')][:40]
    print('| %s | %d | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.2f | %.2f | %.2f | %.2f | %.2f | %.1f |' % (
        nm, e['grid'], wc / 1e6, 100 * g('SQ_WAIT_ANY') / wc, 100 * g('SQ_WAIT_INST_ANY') / wc, 100 * g('SQ_ACTIVE_INST_ANY') / wc,
        100 * g('SQ_VALU_MFMA_BUSY_CYCLES') / (MFMA_NORM * (g('GRBM_GUI_ACTIVE') or 1)), 100 * g('SQ_LDS_IDX_ACTIVE') / (g('SQ_BUSY_CYCLES') or 1),
        100 * g('SQ_LDS_BANK_CONFLICT') / (g('SQ_LDS_IDX_ACTIVE') or 1), g('SQ_INSTS_VALU') / mf, g('SQ_INSTS_SALU') / mf, g('SQ_INSTS_LDS') / mf,
        g('SQ_INSTS_VMEM_RD') / mf, 100 * g('SQ_WAIT_INST_LDS') / wc))
