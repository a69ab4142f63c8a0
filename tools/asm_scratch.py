#!/usr/bin/env python3
"""Where a kernel's scratch (spill) traffic and loops sit in its ISA: python tools/asm_scratch.py file.s mangled_name ..."""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
for name in sys.argv[2:]:
    s = [i for i, l in enumerate(lines) if l.startswith(name + ':')][0]
    e = next(i for i in range(s, len(lines)) if lines[i].startswith('.Lfunc_end'))
    body = lines[s:e]
    inst = [l for l in body if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    print(name, 'instructions', len(inst), 'scratch ops', sum('scratch_' in l for l in inst))
    # basic blocks: label -> (first line, instruction count, scratch ops, loads); back edges mark loops
    blocks, cur = {}, None
    order = []
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            cur = m.group(1)
            blocks[cur] = dict(at=i, n=0, scratch=0, gload=0, lds=0, br=[])
            order.append(cur)
        elif cur and l.startswith('\t') and not l.strip().startswith(('.', ';')):
            b = blocks[cur]
            b['n'] += 1
            b['scratch'] += 'scratch_' in l
            b['gload'] += 'global_load' in l
            b['lds'] += ('ds_' in l)
            m2 = re.search(r's_c?branch\S*\s+(\.LBB\d+_\d+)', l)
            if m2:
                b['br'].append(m2.group(1))
    idx = {k: n for n, k in enumerate(order)}
    loops = []
    for k in order:
        for t in blocks[k]['br']:
            if t in idx and idx[t] <= idx[k]:
                span = order[idx[t]:idx[k] + 1]
                loops.append((t, k, sum(blocks[x]['n'] for x in span), sum(blocks[x]['scratch'] for x in span),
                              sum(blocks[x]['gload'] for x in span), sum(blocks[x]['lds'] for x in span)))
    print('  loops (head, tail, instructions, scratch ops, global loads, LDS ops):')
    for l in sorted(loops, key=lambda x: x[2]):
        print('   ', l)
