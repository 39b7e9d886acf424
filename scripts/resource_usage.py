"""Per-instantiation register / scratch / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` stderr (argv[1])."""
import re
import sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name = b.split()[0]
    if flt not in name:
        continue
    g = lambda k: re.search(k + r': (\d+)', b).group(1)
    print('%-100s V %3s A %3s scratch %3s occ %s vspill %s sspill %s' % (name[:100], g(' VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'),
          g(r'Occupancy \[waves/SIMD\]'), g('VGPRs Spill'), g('SGPRs Spill')))
