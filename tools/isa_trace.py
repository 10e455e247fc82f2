"""Condensed trace of a kernel's biggest loop from hipcc -S output: VALU runs are counted,
memory / LDS / barrier / waitcnt instructions are listed.  usage: isa_trace.py api.s <mangled-prefix>"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith(name) and l.rstrip().endswith(name.split(':')[0] + ':') or l.startswith(name + ':')][0]
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
best = None
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        if best is None or i - a > best[1] - best[0]:
            best = (a, i)
a, b = best
out, run = [], 0
for l in body[a:b + 1]:
    s = l.strip()
    if not l.startswith('\t') or s.startswith(';') or s.startswith('.'):
        if s.startswith('.LBB'):
            if run:
                out.append(f'   [{run} valu]')
                run = 0
            out.append(s.split()[0])
        continue
    op = s.split()[0]
    if op.startswith('v_') or op == 's_nop':
        run += 1
        continue
    if op.startswith(('global_', 'ds_', 's_barrier', 's_waitcnt', 's_cbranch', 's_branch', 'scratch_', 'buffer_')):
        if run:
            out.append(f'   [{run} valu]')
            run = 0
        out.append(s[:72])
res, prev, cnt = [], None, 0
for o in out:
    key = o.split()[0]
    if key == prev and not o.startswith('   [') and key.startswith(('global_', 'ds_')):
        cnt += 1
        continue
    if cnt:
        res[-1] += f'  (x{cnt + 1})'
    res.append(o)
    prev, cnt = key, 0
print('\n'.join(res))
