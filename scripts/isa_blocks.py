#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a gfx950 .s file (development aid).
usage: isa_blocks.py file.s <kernel-name-substring> [min_valu]"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
min_valu = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and key in l and l.split(':')[0].endswith('E'))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks, cur = [], ['entry', []]
for l in lines[start + 1:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append(cur); cur = [m.group(1), []]
    else:
        t = l.strip()
        if t and not t.startswith(';') and not t.startswith('.'):
            cur[1].append(t)
blocks.append(cur)
tot = dict(valu=0, scratch=0)
for name, ins in blocks:
    valu = sum(1 for i in ins if i.startswith('v_'))
    f64 = sum(1 for i in ins if re.match(r'v_(fma|mul|add|max|min|rcp|rsq|sqrt|cmp\w*|ldexp|cvt|frexp\w*|div\w*|trig\w*|rndne|floor|fract)_\w*f64', i))
    fma = sum(1 for i in ins if i.startswith('v_fma_f64'))
    mul = sum(1 for i in ins if i.startswith('v_mul_f64'))
    add = sum(1 for i in ins if i.startswith('v_add_f64'))
    trans = sum(1 for i in ins if re.match(r'v_(rcp|rsq|sqrt)_f64', i))
    scr = sum(1 for i in ins if i.startswith('scratch_'))
    rw = sum(1 for i in ins if i.startswith('v_readlane') or i.startswith('v_writelane'))
    glob = sum(1 for i in ins if i.startswith('global_') or i.startswith('flat_'))
    ds = sum(1 for i in ins if i.startswith('ds_'))
    if valu >= min_valu or scr:
        print(f'{name:12s} n={len(ins):5d} valu={valu:4d} f64={f64:4d} fma={fma:4d} mul={mul:4d} add={add:3d} trans={trans:2d} scratch={scr:3d} lanespill={rw:3d} glob={glob:3d} ds={ds:3d}')
