#!/usr/bin/env python3
"""Instruction mix of the largest basic block (the main loop) of the kernels in a gfx950 .s file whose mangled name
contains every given substring.  usage: loopmix.py file.s substr [substr...]"""
import collections, re, sys
txt = open(sys.argv[1]).read()
keys = sys.argv[2:]
for m in re.finditer(r'^(_Z\S+): +; @\S+\n(.*?)\n\s*s_endpgm', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if not all(k in name for k in keys):
        continue
    blocks = re.split(r'\n\.LBB\S+:[^\n]*', body)
    big = max(blocks, key=lambda b: b.count('\n'))
    ins = [l.strip().split()[0] for l in big.split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
    c = collections.Counter('valu' if i.startswith('v_') else 'salu' if i.startswith('s_') else 'vmem' if i.split('_')[0] in ('global', 'buffer', 'flat', 'scratch') else 'lds' if i.startswith('ds_') else 'other' for i in ins)
    print(name[:90], 'main block:', len(ins), dict(c))
    print('    ', collections.Counter(i for i in ins if i.startswith('v_')).most_common(14))
