"""Instruction mix of the hottest loop of one kernel in a hipcc -S dump: the span between the backward branch that encloses the
most v_mfma (or, without MFMAs, the most VALU) instructions and its target label.
python tests/isa_loop.py file.s <mangled-name-prefix>"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r'^(' + re.escape(name) + r'\S*):.*\n', s, re.M)
body = s[m.end():]
end = re.search(r'^\.Lfunc_end\d+:', body, re.M)
body = body[:end.start()] if end else body
lines = [l.split(';')[0].strip() for l in body.split('\n')]
lines = [l for l in lines if l]
labels = {l[:-1]: i for i, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:$', l)}
best = None
for i, l in enumerate(lines):
    mm = re.match(r'^s_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.match(r'^s_branch\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        span = [x for x in lines[labels[mm.group(1)]:i + 1] if not x.startswith('.')]
        score = (sum(x.startswith('v_mfma') for x in span), sum(x.startswith('v_') for x in span))
        if best is None or score > best[0]:
            best = (score, span, mm.group(1))
score, span, lab = best
c = Counter(x.split()[0] for x in span)
grp = lambda p: sum(n for k, n in c.items() if k.startswith(p))
print(m.group(1)[:60], 'loop', lab, 'instructions', len(span), 'mfma', grp('v_mfma'), 'other valu', grp('v_') - grp('v_mfma'),
      'salu', grp('s_'), 'vmem', grp('global_') + grp('buffer_'), 'ds', grp('ds_'))
print('  ', [(k, n) for k, n in c.most_common(30)])
