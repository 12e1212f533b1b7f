"""count instruction classes per basic block of one kernel in a hipcc -S dump: isa_count.py file.s mangled-substring"""
import re, sys
from collections import Counter
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().endswith(('@' + l.split(':')[0],)) or (l.startswith('_Z') and key in l and ':' in l and not l.startswith('\t')))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i] and i > start + 50)
blocks, cur, name = [], Counter(), 'entry'
def cls(op):
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_load', 'buffer_load', 'flat_load')): return 'vload'
    if op.startswith(('global_store', 'buffer_store', 'flat_store')): return 'vstore'
    if op.startswith('scratch_'): return 'scratch'
    return 'other'
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', t) or (t.startswith('; %bb.') and ':' in t):
        blocks.append((name, cur)); cur = Counter(); name = t.split(':')[0]; continue
    if not t or t.startswith((';', '//', '.')):
        continue
    if False:
        blocks.append((name, cur)); cur = Counter(); name = t.split(':')[0]; continue
    cur[cls(t.split()[0])] += 1
blocks.append((name, cur))
tot = Counter()
for n, c in blocks:
    tot.update(c)
    if sum(c.values()) >= int(sys.argv[3]) if len(sys.argv) > 3 else 40:
        print('%-14s' % n, ' '.join('%s=%d' % kv for kv in sorted(c.items())))
print('TOTAL', dict(tot))
