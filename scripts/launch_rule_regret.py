"""How far the forward / data-gradient launch rule (conv_igemm.hip launch_conv) is from the best forced alternative, per layer shape: runs
scripts/pp_sweep.py for every (batch, input size) of CASES and prints per-tap | RULE | forced stream-K | forced whole tiles with the regret
RULE / best - 1.  usage: [CASES=8:320,8:480,8:608,4:416,8:416,16:416,32:416] python scripts/launch_rule_regret.py   (one gpurun call)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [tuple(int(v) for v in c.split(':')) for c in os.environ.get('CASES', '8:320,8:480,8:608,4:416,8:416,16:416,32:416').split(',')]
worst, n, within4 = [], 0, 0
for B, SIZE in CASES:
    env = dict(os.environ, B=str(B), SIZE=str(SIZE))
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'pp_sweep.py')], env=env, capture_output=True, text=True, timeout=600).stdout
    print('=== input %d x %d, batch %d: forward / data-gradient variants (per-tap | RULE | forced stream-K | forced whole tiles)' % (SIZE, SIZE, B))
    for line in out.splitlines():
        m = re.match(r'(conv\d+)\s+(\S+)\s+(.*)', line)
        if not m:
            continue
        cols = [c.strip() for c in m.group(3).split('   ') if c.strip()]
        if len(cols) != 4 or any(c.startswith('ERR') or c == '-' for c in cols):
            print(line)
            continue
        t = [float(c.split('|')[0]) for c in cols]
        plan = cols[1].split()[1]
        best = min(t)
        regret = t[1] / best - 1.0
        n += 1
        within4 += regret <= 0.04
        worst.append((regret, SIZE, B, m.group(1), m.group(2)))
        print('%-7s %-10s per-tap %6.1f  RULE %6.1f (%s)  sk %6.1f  tile %6.1f   regret %+5.1f%%' % (m.group(1), m.group(2), t[0], t[1], plan, t[2], t[3], 100 * regret), flush=True)
worst.sort(reverse=True)
print('# %d of %d cases within 4 %% of the best column; worst: %s' % (within4, n, ', '.join('%+.1f%% (%s %s at %d, batch %d)' % (100 * w[0], w[3], w[4], w[1], w[2]) for w in worst[:6])))
