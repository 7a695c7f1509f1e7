"""Static instruction mix of one kernel in a hipcc -S dump: python tests/isa_mix.py file.s <mangled-name-prefix>"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
for name in sys.argv[2:]:
    m = re.search(r'^(' + re.escape(name) + r'\S*):.*\n', s, re.M)
    if not m:
        print('not found', name); continue
    body = s[m.end():]
    body = body[:body.find('s_endpgm')]
    lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(('.', ';'))]
    c = Counter(l.split()[0] for l in lines if not l.split(';')[0].strip().endswith(':'))
    grp = lambda p: sum(n for k, n in c.items() if k.startswith(p))
    print(m.group(1)[:40], 'total', sum(c.values()), 'valu', grp('v_'), 'pk', grp('v_pk'), 'ds', grp('ds_'),
          'salu', grp('s_'), 'vmem', grp('global_') + grp('buffer_'))
    print('  ', c.most_common(24))
