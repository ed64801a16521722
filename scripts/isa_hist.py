"""Instruction histogram of one kernel in a hipcc -S listing (development aid)."""
import re, sys
from collections import Counter
path, pat = sys.argv[1], sys.argv[2]
s = open(path).read()
names = sorted(set(re.findall(r'^(_Z\w+):', s, re.M)))
for n in names:
    if not re.search(pat, n): continue
    i = s.index("\n" + n + ":")
    j = s.index(".Lfunc_end", i)
    body = s[i:j]
    lines = [l.strip() for l in body.split('\n')]
    ins = [l.split()[0] for l in lines if l and not l.startswith(('.', ';', '//', '_Z')) and not l.endswith(':')]
    c = Counter(ins)
    print(n[:90], "total", len(ins))
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    print("  VALU", valu, " SALU", sum(v for k, v in c.items() if k.startswith('s_')), " DS", sum(v for k, v in c.items() if k.startswith('ds_')),
          " VMEM", sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'flat_', 'scratch_'))))
    print("  ", ", ".join(f"{k}:{v}" for k, v in c.most_common(40)))
