#!/usr/bin/env python3
"""Dev tool: static instruction mix of one kernel in a hipcc -S listing.  usage: isa_mix.py file.s mangled-name-substring [top]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
m = re.search(r'^(_Z[^\n]*%s[^\n]*):.*?\n(.*?)\.Lfunc_end' % re.escape(sys.argv[2]), s, re.S | re.M)
c = Counter()
for line in m.group(2).split('\n'):
    t = line.strip().split()
    if t and re.match(r'(v_|s_|ds_|global_|buffer_|flat_)', t[0]):
        c[t[0]] += 1
tot = lambda p: sum(v for k, v in c.items() if k.startswith(p))
print(f"VALU {tot('v_')} SALU {tot('s_')} DS {tot('ds_')} VMEM {tot('global_') + tot('buffer_') + tot('flat_')}")
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
    print(f"{v:6d} {k}")
