#!/usr/bin/env python
"""Per-kernel static summary of a `hipcc -S --cuda-device-only` dump: registers, instruction classes.
usage: isa_summary.py file.s <substring of the demangled name> [...]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
labels = [(m.start(), m.group(1)) for m in re.finditer(r'^(_Z\w+):', s, re.M)]
dem = subprocess.run(['c++filt'], input='\n'.join(n for _, n in labels), capture_output=True, text=True).stdout.split('\n')
vg = {m.group(1): (m.group(2), m.group(3)) for m in re.finditer(r'\.amdhsa_kernel (_Z\w+)\n(?:.*\n)*?\s+\.amdhsa_next_free_vgpr (\d+)\n(?:.*\n)*?\s+\.amdhsa_accum_offset (\d+)', s)}
scr = {m.group(1): m.group(2) for m in re.finditer(r'\.amdhsa_kernel (_Z\w+)\n(?:.*\n)*?\s+\.amdhsa_private_segment_fixed_size (\d+)', s)}
for (pos, n), d in zip(labels, dem):
    if not any(f in d for f in sys.argv[2:]):
        continue
    body = s[pos:]
    body = body[:body.find('s_endpgm')]
    ins = [l.split()[0] for l in body.split('\n') if re.match(r'\s+[a-z]', l)]
    cnt = lambda *p: sum(1 for i in ins if i.startswith(p))
    print(f"{d.replace('void ', '')[:64]:64s} vgpr={vg.get(n, ('?', '?'))[0]:>3s} scratch={scr.get(n, '?'):>4s} ins={len(ins):5d} valu={cnt('v_'):5d} "
          f"pk={cnt('v_pk'):4d} mov/xor={cnt('v_mov', 'v_xor'):4d} xlane={sum(1 for l in body.split(chr(10)) if 'dpp' in l or 'permlane' in l):4d} "
          f"ds={cnt('ds_'):4d} vmem={cnt('global_', 'buffer_'):4d} salu={cnt('s_'):4d}")
