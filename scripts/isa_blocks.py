#!/usr/bin/env python3
"""Basic-block instruction statistics of one kernel in a hipcc -S listing (stdin or file): per block the instruction count,
VALU / v_mov / LDS / global counts and the closing branches.  Usage: isa_blocks.py listing.s [min_instructions]"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
mn = int(sys.argv[2]) if len(sys.argv) > 2 else 8
blocks = []; cur = None
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB[0-9_]+):', l)
    if m: cur = [m.group(1), i, collections.Counter(), []]; blocks.append(cur); continue
    if cur is None: cur = ['entry', i, collections.Counter(), []]; blocks.append(cur)
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    cur[2][t.split()[0]] += 1; cur[3].append(t)
for b in blocks:
    c = b[2]; n = sum(c.values())
    if n < mn: continue
    f = lambda p: sum(k for o, k in c.items() if o.startswith(p))
    br = [t for t in b[3] if t.startswith('s_cbranch') or t.startswith('s_branch')]
    print('%-10s line %5d  n %4d  valu %4d  mov %3d  f64 %3d  ds %3d  gl %3d  %s' % (b[0], b[1], n, f('v_'), f('v_mov'), sum(k for o, k in c.items() if 'f64' in o), f('ds_'), f('global_'), ' '.join(x.split()[-1] for x in br[-2:])))
