#!/usr/bin/env python
"""Condensed instruction stream of one kernel from a hipcc -save-temps .s file (MFMA / ds_read / DMA / barrier / waitcnt runs).
usage: isa_stream.py file.s mangled-name-substring [max chars]"""
import re, sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
names = [m.group(1) for m in re.finditer(r'^(\S+):\s*; @', s, re.M) if key in m.group(1)]
if not names:
    names = [m.group(1) for m in re.finditer(r'^(_Z\S+):', s, re.M) if key in m.group(1)]
name = names[0]
i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
out = []
for l in s[i:j].split('\n'):
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        if t.startswith('.LBB'): out.append('\n' + t.split(':')[0] + ':')
        continue
    op = t.split()[0]
    if op.startswith('v_mfma'): k = 'MFMA'
    elif op.startswith('ds_read'): k = 'dsr'
    elif op.startswith('ds_write'): k = 'dsw'
    elif op.startswith('global_load_lds'): k = 'DMA'
    elif op.startswith('global_store'): k = 'gst'
    elif op.startswith('global_load'): k = 'gld'
    elif op.startswith('scratch_'): k = 'SCRATCH'
    elif op.startswith('s_load'): k = 'sld'
    elif op == 's_barrier': k = 'BARRIER'
    elif op == 's_waitcnt': k = 'wait(' + t.split(None, 1)[1] + ')'
    elif op == 's_setprio': k = 'prio' + t.split()[1]
    elif op.startswith('s_cbranch') or op == 's_branch': k = op[2:] + '->' + t.split()[1]
    elif op.startswith('v_'): k = 'v'
    elif op.startswith('s_'): k = 's'
    else: k = op
    out.append(k)
res = []; prev = None; c = 0
for k in out:
    if k == prev: c += 1
    else:
        if prev is not None: res.append(prev + (f'x{c}' if c > 1 else ''))
        prev = k; c = 1
res.append(prev + (f'x{c}' if c > 1 else ''))
txt = ' '.join(res)
print(name); print(txt[:int(sys.argv[3]) if len(sys.argv) > 3 else 20000])
