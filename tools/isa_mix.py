#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing: tools/isa_mix.py file.s <substring of mangled name>"""
import re, sys, collections
s = open(sys.argv[1]).read()
key = sys.argv[2]
for m in re.finditer(r'^(\S*' + re.escape(key) + r'\S*):[^\n]*\n(.*?)\n\.Lfunc_end', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    c = collections.Counter()
    for line in body.split('\n'):
        if not line.startswith('\t'): continue
        t = line.strip().split(' ')[0].split('\t')[0]
        if not t or t.startswith('.') or t.startswith(';'): continue
        c[t] += 1
    groups = collections.Counter()
    for k, v in c.items():
        g = ('global_load' if k.startswith('global_load') else 'global_store' if k.startswith('global_store') else 's_load' if k.startswith('s_load')
             else 'dpp' if 'dpp' in k else 'f64' if k.endswith('_f64') else 'v_other' if k.startswith('v_') else 's_waitcnt' if k == 's_waitcnt' else 's_other' if k.startswith('s_') else k)
        groups[g] += v
    print(name[:100], 'total', sum(c.values()))
    print('  ', dict(groups))
    print('  ', [l.strip() for l in body.split('\n') if 'global_load' in l][:4])
    print('  ', c.most_common(16))
